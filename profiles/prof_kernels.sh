REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof; mkdir -p /tmp/prof gpurun_out/keep
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/o1 2> /tmp/e1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof/kt/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.reader(open(f)))
for r in rows[1:]:
    if 'kslam' in r[0]:
        name=r[0].replace('(anonymous namespace)::','').replace('kslam::','').replace('void ','').split('(')[0]
        print(name.ljust(34), r[1].rjust(5), 'tot_ms=%9.2f'%(float(r[2])/1e6), 'avg_us=%10.1f'%(float(r[3])/1e3), 'max_us=%10.1f'%(float(r[6])/1e3))
PY
