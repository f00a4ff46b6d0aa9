#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q > $O/try.log 2>&1; echo "pytest rc=$?"; tail -3 $O/try.log
KSLAM_SWEEP_ROOM=0 timeout 600 python tools/soak.py 120 33000 > $O/try_soak_room0.txt 2>&1; echo "soak room0 rc=$?"; tail -1 $O/try_soak_room0.txt
timeout 600 python tools/soak.py 120 34000 > $O/try_soak.txt 2>&1; echo "soak rc=$?"; tail -1 $O/try_soak.txt
bash tools/r06_ab_r05.sh 2>&1 | tail -6 | cut -c1-200
bash tools/kprof.sh 2>&1 | grep "k_extract_filter\|^2"
