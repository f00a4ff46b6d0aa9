REPO="$(pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO"; rm -rf /tmp/sprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sprof -o x -- python3 bench.py --strong --steps 3 --warmup 1 --no-cpu-baseline > /tmp/sprof.json 2>/tmp/sprof.err
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/sprof/**/*kernel_stats.csv', recursive=True)[0])))
def clean(n): return n.replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
tot=0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    if 'kslam' in r['Name'] and float(r['TotalDurationNs']) > 2e6:
        print("%-40s calls %5s  avg %9.3f ms  total %9.2f ms" % (clean(r['Name'])[:40], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6))
PY
