#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/full_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/full_gpu.log
python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -2
