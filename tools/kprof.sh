# per-kernel time of one bench run: bash tools/kprof.sh [extra bench args]
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; mkdir -p gpurun_out; rm -rf /tmp/kprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof -o x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline "$@" > /tmp/kprof.json 2>/tmp/kprof.err
python3 - <<'PY'
import csv, glob, json
rows = list(csv.DictReader(open(glob.glob('/tmp/kprof/**/*kernel_stats.csv', recursive=True)[0])))
def clean(n): return n.replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
for r in rows:
    if 'kslam' in r['Name'] and float(r['TotalDurationNs']) > 3e5:
        print("%-34s calls %4s  avg %9.3f ms  total/step %8.3f ms" % (clean(r['Name'])[:34], r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6 / 4))
j = json.load(open('/tmp/kprof.json')); print(j['hot_path']['ms_per_step'], j['hot_path']['phases_ms'])
PY
