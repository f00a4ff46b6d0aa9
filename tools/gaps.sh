# idle time between consecutive kernels of one bench step (where the GPU waits for the host)
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/gp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/gp.json 2>/tmp/gp.err
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/gp/**/*kernel_trace.csv', recursive=True)[0])))
def clean(n): return n.replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), clean(r['Kernel_Name'])) for r in rows if 'kslam' in r['Kernel_Name']))
# last alignment call = from the last k_extract_filter (first kernel of a call) to the end
starts = [i for i, k in enumerate(ks) if k[2].startswith('k_extract_filter')]
i0 = starts[-1]
step = ks[i0:]
busy = sum(e - s for s, e, _ in step); span = step[-1][1] - step[0][0]
print("last step: %d kernels, span %.3f ms, busy %.3f ms, idle %.3f ms" % (len(step), span / 1e6, busy / 1e6, (span - busy) / 1e6))
gaps = []
for a, b in zip(step, step[1:]):
    g = b[0] - a[1]
    gaps.append((g, a[2], b[2]))
big = sorted(gaps, reverse=True)[:25]
for g, a, b in big: print("%8.1f us  after %-34s before %s" % (g / 1e3, a[:34], b[:34]))
print("gaps > 20 us: %d totalling %.3f ms; gaps <= 20 us: %d totalling %.3f ms" % (
    sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6,
    sum(1 for g in gaps if g[0] <= 20000), sum(g[0] for g in gaps if g[0] <= 20000) / 1e6))
PY
