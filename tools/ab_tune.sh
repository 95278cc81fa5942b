for t in 0 1 2 4 8 0; do
  DEEPBINNER_TUNE=$t timeout 300 python bench.py --steps 100 --warmup 10 --no-side-rates --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('tune', $t, round(d['value']), d['roofline']['avg_launch_ms'])"
done
