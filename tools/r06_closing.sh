#!/bin/bash
# round 6 closing runs: GPU suite, profiles (kernel stats + PMC), the bench lines, VALU counters, soak
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
R=${1:-r06}
mkdir -p gpurun_out/keep gpurun_out/r06
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/r06/${R}_full_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/r06/${R}_full_gpu.log
bash profiles/run_profiles.sh $R 2>&1 | tail -8
bash tools/pmc_valu.sh $R 2>&1 | tail -4
bash tools/final_runs.sh ${R}z 2>&1 | tail -9
if [ -z "$NO_SOAK" ]; then timeout 900 python tools/soak.py 300 61000 > gpurun_out/keep/${R}_soak.txt 2>&1; tail -1 gpurun_out/keep/${R}_soak.txt; fi
ls gpurun_out/keep | grep "^$R" | head -40
