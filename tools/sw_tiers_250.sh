#!/bin/bash
# profiles/<R>_sw_hist_250bp.json: candidates and kernel time per SW tier at 250 bp (one 1 M-pair batch of configs[4]'s shape)
R=${1:-r05}
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof250; mkdir -p gpurun_out/keep
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof250 -o x -- python3 bench.py --config 4 --pairs 1000000 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-abi-path > gpurun_out/keep/${R}_bench_250bp_under_rocprof.json 2>/tmp/e250
cp "$(find /tmp/prof250 -name '*kernel_stats.csv' | head -1)" /tmp/kernel_stats_250.csv; grep -i "kslam\|Name" /tmp/kernel_stats_250.csv > gpurun_out/keep/${R}_kernel_stats_250bp.csv
KERNEL_STATS=/tmp/kernel_stats_250.csv READ_LEN=250 PAIRS=1000000 python3 tools/sw_tiers.py > gpurun_out/keep/${R}_sw_hist_250bp.json
READ_LEN=150 PAIRS=1000000 python3 tools/sw_tiers.py > gpurun_out/keep/${R}_sw_hist_150bp.json
python3 bench.py --config 4 --pairs 1000000 --steps 5 --warmup 1 --no-e2e --no-cpu-baseline --no-abi-path > gpurun_out/keep/${R}_bench_250bp.json 2>/dev/null
tail -c 1500 gpurun_out/keep/${R}_sw_hist_250bp.json
