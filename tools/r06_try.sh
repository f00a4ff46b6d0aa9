#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -k "index_build or filter_built or scale or truth or golden" > $O/try.log 2>&1; echo "pytest rc=$?"; tail -3 $O/try.log
bash tools/kprof.sh 2>&1 | grep "setup\|k_extract \|k_split\|k_filter\|k_bucket\|^2"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-pipeline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); i=j['roofline']['index_sort']; print({k:i[k] for k in ('passes','ms','frac','pmc_frac','index_build_ms')})"
