for mb in 0 180 96 45; do echo "== HSSFSST_CHUNK_MB=$mb"; HSSFSST_CHUNK_MB=$mb python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['normalize_avg_launch_ms'])
"; done
