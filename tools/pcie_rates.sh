for v in "--input resident" "--input host32" "--input host16" "--input host32 --prefetch" "--input host16 --prefetch"; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline $v 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], round(d['value']))"
done
