for v in "X=1" "KSLAM_STREAM_DEPTH=4" "KSLAM_LANES=3 KSLAM_STREAM_DEPTH=4" "KSLAM_LANES=3 KSLAM_STREAM_DEPTH=5" "X=1"; do
  env $v python bench.py --no-cpu-baseline --no-abi-path > /tmp/ab.json 2>/dev/null
  python -c "
import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);e=d['e2e'];print('$v', d['value'], e['repetitions_ms_per_step'], 'null', d['e2e_sam_to_dev_null']['ms_per_step'], 'pa', d['e2e_with_pseudo_assembly']['ms_per_step'])"
done
