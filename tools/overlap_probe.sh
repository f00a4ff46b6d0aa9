#!/bin/bash
# Which kernels run while k_extract_filter runs in the pipelined legs?  bash tools/overlap_probe.sh
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof_ov
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -o x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline > /tmp/o_ov 2> /tmp/e_ov
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_ov/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def clean(n):
    return n.replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), clean(r['Kernel_Name']), r.get('Queue_Id', ''), r.get('Stream_Id', '')) for r in rows if 'kslam' in r['Kernel_Name']]
ev.sort()
t0 = ev[0][0]
ex = [e for e in ev if e[2] == 'k_extract_filter']
print('extract_filter dispatches:', len(ex))
for s, e, n, q, st in ex[-14:]:
    over = [(x[2], round((min(e, x[1]) - max(s, x[0])) / 1e3)) for x in ev if x[0] < e and x[1] > s and x[2] != 'k_extract_filter']
    print('t=%9.2f ms dur %8.1f us queue %s stream %s overlapping: %s' % ((s - t0) / 1e6, (e - s) / 1e3, q, st, over[:8]))
PY
