#!/bin/bash
# Regenerates the committed profile summaries on the GPU box: usage  bash profiles/run_profiles.sh rNN
# (kernel-trace/stats and the two PMC passes are separate runs, as the pool requires)
R=${1:-r01}
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof; mkdir -p /tmp/prof gpurun_out/keep
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > gpurun_out/keep/${R}_bench_under_rocprof.json 2> /tmp/e1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/pf -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/o2 2> /tmp/e2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof/pw -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/o3 2> /tmp/e3
python3 - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
def clean(n):
    return n.replace('(anonymous namespace)::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
rows = list(csv.reader(open(glob.glob('/tmp/prof/kt/**/*kernel_stats.csv', recursive=True)[0])))
with open('gpurun_out/keep/%s_kernel_stats_kslam.csv' % R, 'w') as fh:
    w = csv.writer(fh); w.writerow(rows[0])
    for r in rows[1:]:
        if 'kslam' in r[0]:
            w.writerow([clean(r[0])] + r[1:])
pmc = {}
for tag, d in (('FETCH_SIZE', 'pf'), ('WRITE_SIZE', 'pw')):
    for f in glob.glob('/tmp/prof/%s/**/*counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'kslam' not in r['Kernel_Name']:
                continue
            k = clean(r['Kernel_Name'])
            a = pmc.setdefault(k, {}).setdefault(r['Counter_Name'], [0, 0.0, 0.0, []])
            v = float(r["Counter_Value"]); a[0] += 1; a[1] += v; a[2] = max(a[2], v); a[3].append(v)
out = {k: {c: {'dispatches': a[0], 'sum': a[1], 'mean': a[1] / a[0], 'max': a[2], 'min': min(a[3]),
               'per_dispatch': a[3] if k.startswith(('k_scatter<4>', 'k_tile_hist<4>', 'k_extract_filter', 'k_join_fill')) else None}
           for c, a in cs.items()} for k, cs in pmc.items()}
json.dump(out, open('gpurun_out/keep/%s_pmc_kslam.json' % R, 'w'), indent=1, sort_keys=True)
for k in sorted(out):
    print(k.ljust(30), {c: (v['dispatches'], '%.4g' % v['mean'], '%.4g' % v['max']) for c, v in out[k].items()})
PY
