for nd in 16 32 48 64; do
  for cfg in 1 4; do
    KSLAM_SW_UNKNOWN_ND=$nd python3 bench.py --config $cfg --pairs 1000000 --steps 5 --warmup 1 --no-e2e --no-cpu-baseline --no-abi-path 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unknown_nd', $nd, 'config', $cfg, d['hot_path']['phases_ms']['ms_sw'], d['hot_path']['ms_per_step'], d['hot_path']['verified']['ok'])"
  done
done
