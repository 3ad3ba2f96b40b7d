#!/usr/bin/env python3
"""Summarise rocprofv3 sqlite outputs (kernel-trace stats + PMC averages per dispatch) as JSON.
usage: rocprof_summary.py out.json kt_results.db [pmc_results.db ...]"""
import json
import sqlite3
import sys

out = {}
db = sqlite3.connect(sys.argv[2])
out["kernel_trace_stats"] = [dict(zip(["name", "calls", "total_us", "avg_us", "pct"], r))
                             for r in db.execute("select * from top_kernels")]
for path in sys.argv[3:]:
    db = sqlite3.connect(path)
    q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
         "group by kernel_name, counter_name")
    for name, ctr, avg, cnt in db.execute(q):
        out.setdefault("pmc_avg_per_dispatch", {}).setdefault(name, {})[ctr] = avg
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["kernel_trace_stats"], indent=1))
