#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
timeout 1200 python -m pytest tests/test_gpu_config1_full.py -x -q -s --durations=5 2>&1 | grep -v "amdgpu.ids" | tail -20
