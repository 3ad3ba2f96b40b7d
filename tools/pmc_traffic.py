#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from a tools/rocprof_summary.py summary (FETCH_SIZE / WRITE_SIZE passes).
usage: pmc_traffic.py summary.json out.json
Counter values are KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950."""
import json, sys
s = json.load(open(sys.argv[1]))["pmc_avg_per_dispatch"]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 5 --warmup 1 "
                 "--settle-steps 0 --no-cpu-baseline` (tools/profile_round.sh)",
       "units": "counter values are KiB; bytes = value * 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
                "half of a streaming read; dword-wide reads here, so the read side is approximate)"}
for name, v in s.items():
    if "FETCH_SIZE" not in v or "rocclr" in name:
        continue
    key = "fsst_core128_kernel" if "core128" in name else name.split("::")[-1].split("(")[0]
    e = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
         "hbm_bytes_per_launch": int(round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024))}
    if "core128" in name:
        e["algorithmic_bytes_per_launch"] = 360000 * 1024
    out[key] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
