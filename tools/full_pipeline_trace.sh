#!/bin/bash
# Kernel times under the full_pipeline legs (FASTQ text -> SAM text):  bash tools/full_pipeline_trace.sh TAG
R=${1:-fp}
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof_fp; mkdir -p gpurun_out/keep
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fp -o x -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline > gpurun_out/keep/${R}_full_pipeline_bench.json 2> /tmp/e_fp
python3 - "$R" <<'PY'
import csv, glob, sys
R = sys.argv[1]
rows = list(csv.reader(open(glob.glob('/tmp/prof_fp/**/*kernel_stats.csv', recursive=True)[0])))
def clean(n):
    return n.replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
with open('gpurun_out/keep/%s_full_pipeline_kernels.txt' % R, 'w') as fh:
    for r in rows[1:]:
        if 'kslam' in r[0]:
            line = '%-34s calls %6s total_ms %10.2f avg_us %10.1f max_us %10.1f' % (clean(r[0]), r[1], float(r[2]) / 1e6, float(r[3]) / 1e3, float(r[6]) / 1e3)
            fh.write(line + '\n'); print(line)
PY
