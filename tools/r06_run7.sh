#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
BA="--steps 5 --warmup 2 --no-cpu-baseline --no-full-pipeline"
KSLAM_DEBUG=1 python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-full-pipeline 2>&1 >/dev/null | grep "SW \|cigar" | sed -n 1,60p
for v in 1 0 1 0; do KSLAM_SW_SWEEP=$v python bench.py $BA > $O/run7_plain.json 2>/dev/null; python3 -c "
import json; j=json.loads(open('gpurun_out/r06/run7_plain.json').read().strip().splitlines()[-1]); print('sweep=$v:', j['hot_path']['phases_ms'], j['hot_path']['verified']['ok'])"; done
python bench.py --read-len 250 $BA > $O/run7_250.json 2>/dev/null; python3 -c "
import json; j=json.loads(open('gpurun_out/r06/run7_250.json').read().strip().splitlines()[-1]); print('250bp:', j['hot_path']['phases_ms'], j['hot_path']['verified']['ok'])"
KSLAM_SW_SWEEP=0 python bench.py --read-len 250 $BA > $O/run7_250.json 2>/dev/null; python3 -c "
import json; j=json.loads(open('gpurun_out/r06/run7_250.json').read().strip().splitlines()[-1]); print('250bp sweep=0:', j['hot_path']['phases_ms'], j['hot_path']['verified']['ok'])"
bash tools/gaps.sh > $O/gaps7.txt 2>&1; head -10 $O/gaps7.txt; tail -1 $O/gaps7.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $O/t7.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t7.log
timeout 600 python tools/soak.py 60 9700 > $O/soak7.txt 2>&1; echo "soak rc=$?"; tail -1 $O/soak7.txt
