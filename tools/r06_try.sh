#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $O/try.log 2>&1; echo "pytest rc=$?"; tail -3 $O/try.log
timeout 600 python tools/soak.py ${SOAK:-80} 12000 > $O/try_soak.txt 2>&1; echo "soak rc=$?"; tail -1 $O/try_soak.txt
bash tools/r06_ab_r05.sh 2>&1 | tail -6 | cut -c1-200
