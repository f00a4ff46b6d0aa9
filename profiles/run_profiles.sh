#!/bin/bash
# Regenerates the committed profile summaries on the GPU box: usage  bash profiles/run_profiles.sh rNN
# (kernel-trace/stats and the two PMC passes are separate runs, as the pool requires)
R=${1:-r01}
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof; mkdir -p /tmp/prof gpurun_out/keep
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-pipeline > gpurun_out/keep/${R}_bench_under_rocprof.json 2> /tmp/e1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/pf -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-pipeline > /tmp/o2 2> /tmp/e2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof/pw -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-pipeline > /tmp/o3 2> /tmp/e3
python3 - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
def clean(n):
    return n.replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
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
# ---- the k-mer sort PHASE (rocprof (FETCH_SIZE + WRITE_SIZE) / t_sort beside the formula value): every dispatch
# between a k_extract_filter and the next k_join_fill whose kernel belongs to the radix sort, summed per alignment call ----
SORT = ('k_tile_hist_bytes', 'k_tile_hist<4>', 'k_chunk_scan', 'k_col_scan', 'k_bin_scan', 'k_scatter<4>')
phase = {}
for tag, d in (('FETCH_SIZE', 'pf'), ('WRITE_SIZE', 'pw')):
    rows = []
    for f in glob.glob('/tmp/prof/%s/**/*counter_collection.csv' % d, recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if 'kslam' in r['Kernel_Name'] and r['Counter_Name'] == tag]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    calls, cur, inside = [], None, False
    for r in rows:
        k = clean(r['Kernel_Name'])
        if k.startswith('k_extract_filter'):
            inside, cur = True, {'kib': 0.0, 'dispatches': 0, 'by_kernel': {}}
        elif k.startswith('k_join_fill'):
            if inside and cur and cur['dispatches']:
                calls.append(cur)
            inside = False
        elif inside and k.startswith(SORT):
            v = float(r['Counter_Value'])
            cur['kib'] += v
            cur['dispatches'] += 1
            cur['by_kernel'][k] = cur['by_kernel'].get(k, 0.0) + v
    phase[tag] = calls
if phase.get('FETCH_SIZE') and phase.get('WRITE_SIZE'):
    n = min(len(phase['FETCH_SIZE']), len(phase['WRITE_SIZE']))
    f = sum(c['kib'] for c in phase['FETCH_SIZE'][:n]) / n
    w = sum(c['kib'] for c in phase['WRITE_SIZE'][:n]) / n
    sp = {'alignment_calls_seen': n, 'dispatches_per_call': phase['FETCH_SIZE'][0]['dispatches'],
          'fetch_kib_per_call_raw': f, 'write_kib_per_call_raw': w,
          'bytes_per_call': int((2 * f + w) * 1024),
          'method': 'all radix-sort dispatches between k_extract_filter and k_join_fill of one alignment call; FETCH_SIZE doubled '
                    '(gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md section HBM), WRITE_SIZE as is; unit KiB',
          'by_kernel_fetch_kib': phase['FETCH_SIZE'][0]['by_kernel'], 'by_kernel_write_kib': phase['WRITE_SIZE'][0]['by_kernel']}
    json.dump(sp, open('gpurun_out/keep/%s_sort_phase_pmc.json' % R, 'w'), indent=1, sort_keys=True)
    print('sort phase:', sp['bytes_per_call'], 'bytes per alignment call over', sp['dispatches_per_call'], 'dispatches')
# ---- the ONE-TIME sort of the genome k-mer records (kslam_set_index: the *_setup kernels; one index build per run) ----
ISORT = ('k_tile_hist_setup', 'k_tile_hist_bytes_setup', 'k_tile_hist_bytes_skew_setup', 'k_scatter_setup')
isum = {}
for tag, d in (('FETCH_SIZE', 'pf'), ('WRITE_SIZE', 'pw')):
    rows = []
    for f in glob.glob('/tmp/prof/%s/**/*counter_collection.csv' % d, recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if 'kslam' in r['Kernel_Name'] and r['Counter_Name'] == tag]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    # the record sort ends where k_split_soa starts (since round 6 the membership filter's probe words go through the same
    # *_setup kernels afterwards: not part of this sum)
    ends = [int(r['Dispatch_Id']) for r in rows if clean(r['Kernel_Name']).startswith(('k_split_soa', 'k_split_tables'))]
    end = ends[0] if ends else 1 << 62
    tot, disp, by = 0.0, 0, {}
    for r in rows:
        k = clean(r['Kernel_Name'])
        if int(r['Dispatch_Id']) < end and k.startswith(ISORT):
            v = float(r['Counter_Value']); tot += v; disp += 1; by[k] = by.get(k, 0.0) + v
    isum[tag] = (tot, disp, by)
index_sort = None
if isum['FETCH_SIZE'][1] and isum['WRITE_SIZE'][1]:
    f, w = isum['FETCH_SIZE'][0], isum['WRITE_SIZE'][0]
    index_sort = {'dispatches': isum['FETCH_SIZE'][1], 'fetch_kib_raw': f, 'write_kib_raw': w, 'bytes': int((2 * f + w) * 1024),
                  'by_kernel_fetch_kib': isum['FETCH_SIZE'][2], 'by_kernel_write_kib': isum['WRITE_SIZE'][2],
                  'method': 'every dispatch of the one-time sort of the genome k-mer records, i.e. before k_split_soa (k_tile_hist_bytes_setup<...>, k_scatter_setup<4>; the three scan '
                            'kernels between them move < 0.1 % and share their names with the per-batch sort) of the run\'s single kslam_set_index; '
                            'FETCH_SIZE doubled (gfx950), WRITE_SIZE as is; unit KiB'}
    json.dump(index_sort, open('gpurun_out/keep/%s_index_sort_pmc.json' % R, 'w'), indent=1, sort_keys=True)
    print('index sort:', index_sort['bytes'], 'bytes over', index_sort['dispatches'], 'dispatches')
# ---- profiles/traffic.json as bench.py reads it (this round's numbers on top; copy it over the committed file) ----
sc = out.get('k_scatter<4>', {})
if 'FETCH_SIZE' in sc and 'WRITE_SIZE' in sc:
    tj = {'kernel': 'k_scatter<4>', 'source': 'profiles/%s_pmc_kslam.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, bench.py --steps 1 --warmup 0)' % R,
          'fetch_kb_per_launch_raw': sc['FETCH_SIZE']['mean'], 'write_kb_per_launch_raw': sc['WRITE_SIZE']['mean'],
          'k_scatter_bytes_per_launch': int((2 * sc['FETCH_SIZE']['mean'] + sc['WRITE_SIZE']['mean']) * 1024),
          'method': 'mean over the dispatches of k_scatter<4>; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md section HBM), WRITE_SIZE as is; counter unit KiB',
          'sort_phase': json.load(open('gpurun_out/keep/%s_sort_phase_pmc.json' % R)) if phase.get('FETCH_SIZE') and phase.get('WRITE_SIZE') else None,
          'index_sort': index_sort}
    json.dump(tj, open('gpurun_out/keep/%s_traffic.json' % R, 'w'), indent=1, sort_keys=True)
for k in sorted(out):
    print(k.ljust(30), {c: (v['dispatches'], '%.4g' % v['mean'], '%.4g' % v['max']) for c, v in out[k].items()})
PY
