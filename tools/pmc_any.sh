#!/bin/bash
# usage: bash tools/pmc_any.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- <command ...>
# Runs <command> once per counter pass under rocprofv3 --pmc and prints the per-kernel average of every counter.
tag=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
i=0
for c in "${passes[@]}"; do
  rocprofv3 --pmc $c -d $O/p$i -o p$i -- "$@" > $O/p$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import sqlite3, json, glob
out = {}
for f in sorted(glob.glob("$O/p*/*_results.db")):
    db = sqlite3.connect(f)
    for name, ctr, avg, cnt in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(name[:90], {})[ctr] = avg
        out[name[:90]]["_dispatches"] = cnt
json.dump(out, open("$O.json", "w"), indent=1)
for k, v in out.items():
    if "hssfsst" in k: print(k, {a: round(b, 1) for a, b in v.items()})
PY
rm -rf $O
