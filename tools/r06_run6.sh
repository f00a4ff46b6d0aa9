#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
BA="--steps 5 --warmup 2 --no-cpu-baseline --no-full-pipeline"
KSLAM_DEBUG=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-pipeline 2>&1 >/dev/null | grep "cigar" | head -30
for i in 1 2; do python bench.py $BA > $O/run6_plain.json 2>/dev/null; python3 -c "
import json; j=json.loads(open('gpurun_out/r06/run6_plain.json').read().strip().splitlines()[-1]); print('no profiler:', j['hot_path']['phases_ms'])"; done
bash tools/gaps.sh > $O/gaps6.txt 2>&1; head -14 $O/gaps6.txt; tail -1 $O/gaps6.txt
