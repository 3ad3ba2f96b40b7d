#!/bin/bash
# usage: bash tools/kt_any.sh <tag> -- <command ...>   kernel-trace stats of any command, summary printed + json kept
tag=$1; shift; shift
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- "$@" > $O/kt.log 2>&1
python - <<PY
import sqlite3, json
db = sqlite3.connect("$O/kt/kt_results.db")
rows = [dict(zip(["name", "calls", "total_us", "avg_us", "pct"], r)) for r in db.execute("select * from top_kernels")]
json.dump(rows, open("$O.json", "w"), indent=1)
for r in rows: print(r["name"][:90], r["calls"], round(r["avg_us"], 2))
PY
tail -2 $O/kt.log | cut -c1-300
rm -rf $O
