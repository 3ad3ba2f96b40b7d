#!/bin/bash
# usage: bash tools/pmc_lib.sh <tag> lib.so [lib2.so ...]  -- per-16-frame-group instruction mix of the team kernel of development builds
# (rocprofv3 --pmc, two passes per build, 8 execs of the C2 workload each; per group = per dispatch / 128000)
tag=$1; shift
R=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $R
for lib in "$@"; do
  O=$R/gpurun_out/$tag.$(basename $lib .so); mkdir -p $O
  T16_OFFSET=${T16_OFFSET:-0} T16_ROUNDS=2 T16_EXECS=4 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/p1 -o p1 -- python tools/t16_probe.py $lib > $O/p1.log 2>&1
  T16_OFFSET=${T16_OFFSET:-0} T16_ROUNDS=2 T16_EXECS=4 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p2 -o p2 -- python tools/t16_probe.py $lib > $O/p2.log 2>&1
  python - <<PY
import sqlite3, glob
out = {}
for f in sorted(glob.glob("$O/p*/*_results.db")):
    db = sqlite3.connect(f)
    for name, ctr, avg in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        if "team16" in name or "teamq" in name: out[ctr] = avg
g = 128000.0
print("$lib", " ".join(f"{k.replace('SQ_', '')}={v / g:.1f}" for k, v in sorted(out.items())))
PY
  rm -rf $O
done
