for v in 0 3051 57 2057 3042 3034; do
  MAPPO_GAE_VARIANT=$v python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant $v  frac', r['frac'], 'launch_ms', r['launch_ms'], 'step', d['ms_per_step'])"
done
