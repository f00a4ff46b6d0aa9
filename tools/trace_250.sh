#!/bin/bash
# Kernel trace of the 250-bp configuration (BASELINE configs[4] shape, 1 M pairs per step):
#   bash tools/trace_250.sh TAG  -> gpurun_out/keep/TAG_bench_250bp.json, TAG_kernel_stats_250bp.txt
R=${1:-r02b}
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof_250; mkdir -p gpurun_out/keep
python3 bench.py --read-len 250 --steps 5 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > gpurun_out/keep/${R}_bench_250bp.json 2> /tmp/e_250a
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_250 -o x -- python3 bench.py --read-len 250 --steps 3 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/o_250 2> /tmp/e_250
python3 - "$R" <<'PY'
import csv, glob, sys
R = sys.argv[1]
rows = list(csv.reader(open(glob.glob('/tmp/prof_250/**/*kernel_stats.csv', recursive=True)[0])))
def clean(n):
    return n.replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
with open('gpurun_out/keep/%s_kernel_stats_250bp.txt' % R, 'w') as fh:
    fh.write('# rocprofv3 --kernel-trace --stats of: bench.py --read-len 250 --steps 3 --warmup 1 (6 timed-or-warm steps + verification runs)\n')
    for r in rows[1:]:
        if 'kslam' in r[0]:
            line = '%-38s calls %6s total_ms %10.2f avg_us %10.1f min_us %10.1f max_us %10.1f' % (clean(r[0]), r[1], float(r[2]) / 1e6, float(r[3]) / 1e3, float(r[5]) / 1e3, float(r[6]) / 1e3)
            fh.write(line + '\n'); print(line)
PY
python3 -c "
import json;d=json.loads(open('gpurun_out/keep/${R}_bench_250bp.json').read().strip().splitlines()[-1])
print(d['hot_path']['reads_per_s'], d['hot_path']['ms_per_step'], d['hot_path']['phases_ms'], d['hot_path']['counts'], d['hot_path']['verified'])"
