for cfg in "1 0" "1 256" "1 512" "1 1024" "4 256" "4 512" "4 128" "2 512" "8 256"; do set -- $cfg; echo "== CHUNKS=$1 ZGRID=$2"; HSSFSST_CHUNKS=$1 HSSFSST_ZGRID=$2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['normalize_avg_launch_ms'])
"; done
