# A/B of an environment switch on ONE box: bash tools/ab_bench.sh VAR=VALUE [rounds]
SW="$1"; N=${2:-2}
for i in $(seq 1 $N); do
  for v in base "$SW"; do
    if [ "$v" = base ]; then python bench.py --no-cpu-baseline --no-abi-path > /tmp/ab.json 2>/dev/null; else env $v python bench.py --no-cpu-baseline --no-abi-path > /tmp/ab.json 2>/dev/null; fi
    python -c "
import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);e=d['e2e'];print('$v', d['value'], e['repetitions_ms_per_step'], 'null', d['e2e_sam_to_dev_null']['ms_per_step'], 'pa', d['e2e_with_pseudo_assembly']['ms_per_step'], 'hot', d['hot_path']['ms_per_step'])"
  done
done
