#!/bin/bash
# usage: bash tools/pmc_quick.sh <tag>  -- instruction-mix PMC pass of a short bench run, summary to gpurun_out/<tag>.json
tag=${1:-pmcq}
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
S="--steps 5 --warmup 1 --settle-steps 0 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM -d $O/p1 -o p1 -- python bench.py $S > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $O/p2 -o p2 -- python bench.py $S > $O/p2.log 2>&1
python - <<PY
import sqlite3, json
out = {}
for f in ("$O/p1/p1_results.db", "$O/p2/p2_results.db"):
    db = sqlite3.connect(f)
    for name, ctr, avg in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(name[:60], {})[ctr] = avg
json.dump(out, open("$O.json", "w"), indent=1)
for k, v in out.items():
    if "core" in k:
        w = v["SQ_WAVES"]
        print(k, {a: round(b / w, 1) for a, b in v.items()})
PY
rm -rf $O
