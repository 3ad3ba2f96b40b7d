#!/bin/bash
# usage: bash tools/kt_quick.sh <tag> -- kernel-trace stats of the default bench (no CPU leg), summary printed
tag=${1:-ktq}
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu-baseline > $O/kt.log 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("$O/kt/kt_results.db")
for r in db.execute("select * from top_kernels"): print(r[0][:70], r[1], round(r[3], 2))
PY
tail -1 $O/kt.log | cut -c1-250
rm -rf $O/kt
