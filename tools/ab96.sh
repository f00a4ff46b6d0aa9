for RL in 150 250; do for V in "" "KSLAM_SW_NO96=1"; do
  env $V KSLAM_DEBUG=1 python bench.py --read-len $RL --steps 6 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/o.json 2> /tmp/e.txt
  echo "RL=$RL $V"; grep "SW tier\|SW full" /tmp/e.txt | head -7 | tr '\n' ';'; echo
  python -c "
import json;d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]);print(d['hot_path']['ms_per_step'], d['hot_path']['phases_ms']['ms_sw'], d['hot_path']['verified']['ok'])"
done; done
