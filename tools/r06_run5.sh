#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $O/t5.log 2>&1; echo "pytest rc=$?"; tail -5 $O/t5.log
timeout 600 python tools/soak.py 80 9500 > $O/soak5.txt 2>&1; echo "soak rc=$?"; tail -2 $O/soak5.txt
BA="--steps 5 --warmup 2 --no-cpu-baseline --no-full-pipeline"
rm -rf /tmp/kp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o x -- python3 bench.py $BA > $O/run5.json 2> /tmp/kp.err
cp $(find /tmp/kp -name '*kernel_stats.csv' | head -1) $O/run5_kernel_stats.csv
python3 - <<'PY'
import json, csv
j = json.loads(open('gpurun_out/r06/run5.json').read().strip().splitlines()[-1])
print(j['hot_path']['phases_ms'], j['hot_path']['verified']['ok'])
i = j['roofline']['index_sort']; print({k: i[k] for k in ('passes', 'ms', 'frac', 'index_build_ms')})
for r in csv.DictReader(open('gpurun_out/r06/run5_kernel_stats.csv')):
    n = r['Name']
    if any(k in n for k in ('hist_bytes_setup', 'k_scatter_setup', 'k_banded', 'k_cigar_sys', 'k_systolic', 'k_finalize')):
        print('   ', n.split('(')[0].split('::')[-1][:60], r['Calls'], round(float(r['AverageNs'])/1e6, 4), 'avg', round(float(r['MinNs'])/1e6, 4), 'min', round(float(r['MaxNs'])/1e6, 4), 'max')
PY
python bench.py $BA > $O/run5_plain.json 2>/dev/null; python3 -c "
import json; j=json.loads(open('gpurun_out/r06/run5_plain.json').read().strip().splitlines()[-1]); print('no profiler:', j['hot_path']['phases_ms'])"
bash tools/gaps.sh > $O/gaps5.txt 2>&1; head -12 $O/gaps5.txt; tail -1 $O/gaps5.txt
