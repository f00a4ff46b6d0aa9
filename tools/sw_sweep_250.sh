for V in "" "KSLAM_SW_UNKNOWN_ND=32" "KSLAM_SW_UNKNOWN_ND=64" "KSLAM_SW_NO48=1"; do
  env $V python bench.py --read-len 250 --steps 4 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/o.json 2> /tmp/e.txt
  python -c "
import json;d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]);print('$V', d['hot_path']['ms_per_step'], d['hot_path']['phases_ms']['ms_sw'], d['hot_path']['verified']['ok'])"
done
