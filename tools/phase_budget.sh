#!/bin/bash
# usage: bash tools/phase_budget.sh <out.json>   (on the GPU box, from the repo root)
# Per-16-frame-group instruction and cycle budget of the three z-score paths of the canonical configuration (C2 shape, no
# extras): rocprofv3 --pmc passes (counters only, separate runs) of `python bench.py --no-extras --no-cpu-baseline` with
# HSSFSST_NO_FUSED / HSSFSST_TEAM_ONLY selecting the path; per-group = per-dispatch / (1024 x 125).
out=${1:-gpurun_out/phase_budget.json}
R=$(pwd); O=$R/gpurun_out/pb_tmp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
S="--steps 5 --warmup 1 --settle-steps 0 --no-cpu-baseline --no-extras"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
for path in fused twolaunch team; do
  case $path in fused) E="";; twolaunch) E="HSSFSST_NO_FUSED=1";; team) E="HSSFSST_TEAM_ONLY=1";; esac
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    env $E rocprofv3 --pmc $P -d $O/$path$i -o p -- python bench.py $S > $O/$path$i.log 2>&1
  done
done
python - <<PY
import sqlite3, json, glob
res = {"how": "rocprofv3 --pmc (three separate passes per path) of python bench.py --no-extras --no-cpu-baseline --steps 5; per 16-frame group = per dispatch / 128000; cycles are summed over waves (SQ_WAVE_CYCLES, SQ_WAIT_*: quad-cycles per wave)", "paths": {}}
for path in ("fused", "twolaunch", "team"):
    k = {}
    for i in (1, 2, 3):
        for f in glob.glob("$O/%s%d/**/*_results.db" % (path, i), recursive=True):
            db = sqlite3.connect(f)
            for name, ctr, avg in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
                if "canon" in name or "team128" in name or "normalize" in name:
                    k.setdefault(name.split("(")[0].replace("void hssfsst::", ""), {})[ctr] = avg
    res["paths"][path] = {kn: {"per_dispatch": {c: round(v) for c, v in d.items()}, "per_group": {c: round(v / 128000.0, 2) for c, v in d.items()}} for kn, d in k.items()}
json.dump(res, open("$out", "w"), indent=1)
for path, ks in res["paths"].items():
    for kn, d in ks.items():
        print(path, kn, d["per_group"])
PY
rm -rf $O
