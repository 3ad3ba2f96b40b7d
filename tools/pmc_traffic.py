#!/usr/bin/env python3
"""profiles/rNN_pmc.json from a tools/rocprof_summary.py summary (separate --pmc passes).
usage: pmc_traffic.py summary.json out.json
Counter values of FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
(it reports half of a wide streaming read).  The file is keyed to the SHA-256 of the kernel sources it was measured
on; bench.py quotes it only for that binary."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
s = json.load(open(sys.argv[1]))
out = {"source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ instruction mix | SQ wait/busy + MFMA | TCC hit/miss, one pass each) of "
                 "`python bench.py --steps 5 --warmup 1 --settle-steps 0 --no-cpu-baseline` (tools/profile_round.sh); averages per dispatch",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 reports half of a wide "
                "streaming read: MI355X_MICROARCH.md); SQ_* are wave-instruction / quad-cycle counts summed over the launch",
       "csrc_sha256": bench.csrc_digest(), "kernels": {}}
for name, v in s.get("pmc_avg_per_dispatch", {}).items():
    if "hssfsst" not in name:
        continue
    e = dict(v)
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        e["hbm_bytes_per_launch"] = int(round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024))
    out["kernels"][name] = e
out["kernel_trace_stats"] = s.get("kernel_trace_stats")
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
