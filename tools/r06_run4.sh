#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "index_build_stats or filter_built or merge_join or golden or extract" > $O/t4.log 2>&1; echo "pytest rc=$?"; tail -5 $O/t4.log
BA="--steps 3 --warmup 1 --no-cpu-baseline --no-full-pipeline"
rm -rf /tmp/kp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o x -- python3 bench.py $BA > $O/index2.json 2> /tmp/kp.err
cp $(find /tmp/kp -name '*kernel_stats.csv' | head -1) $O/index2_kernel_stats.csv
python3 - <<'PY'
import json, csv
j = json.loads(open('gpurun_out/r06/index2.json').read().strip().splitlines()[-1])
print(j['hot_path']['phases_ms'], j['roofline']['index_sort'], j['hot_path']['verified']['ok'])
for r in csv.DictReader(open('gpurun_out/r06/index2_kernel_stats.csv')):
    n = r['Name']
    if any(k in n for k in ('k_filter', 'setup', 'k_split', 'k_bucket', 'k_extract<', 'k_extract(')):
        print('   ', n.split('(')[0].split('::')[-1][:40], r['Calls'], round(float(r['AverageNs'])/1e6, 4), 'avg', round(float(r['MinNs'])/1e6, 4), 'min', round(float(r['MaxNs'])/1e6, 4), 'max', round(float(r['TotalDurationNs'])/1e6, 3), 'total')
PY
