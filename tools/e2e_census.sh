# Per-batch GPU time of every kernel inside the e2e legs (the windows between the first and last k_sam_write):  bash tools/e2e_census.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof/kt -o x -- python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --strong-n1 off $BENCH_ARGS > /tmp/o1 2> /tmp/e1
python3 - <<'PY'
import csv, glob, re, collections
rows = list(csv.DictReader(open(glob.glob('/tmp/prof/kt/**/*kernel_trace.csv', recursive=True)[0])))
ev = []
for r in rows:
    m = re.search(r'(k_[A-Za-z0-9_]+(<[^>]*>)?)', r['Kernel_Name'])
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:40]))
ev.sort()
sw = [e for e in ev if e[2].startswith('k_sam_write')]
t0, t1 = sw[0][0] - 60_000_000, sw[-1][1]
win = [e for e in ev if t0 <= e[0] <= t1]
nb = sum(1 for e in win if e[2] == 'k_join_fill')
tot = collections.defaultdict(float); cnt = collections.Counter()
for s, e, n in win: tot[n] += (e - s) / 1e6; cnt[n] += 1
# union of busy time (kernels of the lanes overlap)
busy, cur_s, cur_e = 0.0, None, None
for s, e, n in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += (cur_e - cur_s) / 1e6
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += (cur_e - cur_s) / 1e6
print('batches in the e2e windows: %d; sum of kernel time per batch %.2f ms; union (GPU busy) per batch %.2f ms; window per batch %.2f ms' % (nb, sum(tot.values()) / nb, busy / nb, (t1 - t0) / 1e6 / nb))
groups = {'text': ('k_sam_', 'k_lca', 'k_per_read'), 'index': ('k_fq', 'k_fastq', 'k_index', 'k_gather'), 'details': ('k_row_det', 'k_md', 'k_detail'), 'pairs': ('k_pair', 'k_screen', 'k_insert', 'k_group')}
import statistics
dur = collections.defaultdict(list)
for s, e, n in win: dur[n].append((e - s) / 1e6)
for n, t in sorted(tot.items(), key=lambda x: -x[1])[:40]:
    d = sorted(dur[n])
    print('%-40s %6.1f calls/batch  %8.3f ms/batch   min %.3f med %.3f p90 %.3f max %.3f' % (n, cnt[n] / nb, t / nb, d[0], d[len(d) // 2], d[int(len(d) * 0.9)], d[-1]))
# what runs at the same time as the slowest dispatches of the first SW tier
sb = sorted((e for e in win if e[2].startswith('k_sw_band<160, 8, 2')), key=lambda e: e[0] - e[1])[:3]
for s0, e0, n0 in sb:
    print('-- %s of %.2f ms overlaps:' % (n0, (e0 - s0) / 1e6))
    for s, e, n in win:
        if s < e0 and e > s0 and (s, e, n) != (s0, e0, n0):
            print('     %-36s %.2f ms, overlap %.2f ms' % (n, (e - s) / 1e6, (min(e, e0) - max(s, s0)) / 1e6))
PY
