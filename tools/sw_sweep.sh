# where gapped candidates start, with / without the 48-diagonal tier (results are identical; times differ)
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$REPO" || exit 1
for cfg in "48 0" "16 0" "32 0" "64 0" "32 1" "64 1"; do
  set -- $cfg
  if [ "$2" = 1 ]; then export KSLAM_SW_NO48=1; else unset KSLAM_SW_NO48; fi
  KSLAM_SW_UNKNOWN_ND=$1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline "${@:3}" > /tmp/sw.json 2>/tmp/sw.err
  python3 -c "
import json,sys; j=json.load(open('/tmp/sw.json')); print('unknown_nd', '$1', 'no48', '$2', 'ms/step', j['hot_path']['ms_per_step'], 'sw', j['hot_path']['phases_ms']['ms_sw'], 'cigar', j['hot_path']['phases_ms']['ms_cigar'], 'ok', j['hot_path']['verified']['ok'])"
done
