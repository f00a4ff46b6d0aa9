for pb in 16 32 64 128 256; do
  KSLAM_PLAN_BLOCKS=$pb python3 bench.py --steps 10 --warmup 2 --no-e2e --no-cpu-baseline --no-abi-path 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plan_blocks', $pb, d['hot_path']['phases_ms']['ms_sw'], d['hot_path']['ms_per_step'])"
done
