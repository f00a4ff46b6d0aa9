# e2e kernel profile (one command per gpurun call): bash tools/prof_sam.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -o x -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --strong-n1 off $BENCH_ARGS > /tmp/o1 2> /tmp/e1
python3 - <<'PY'
import csv, glob, re
rows=list(csv.reader(open(glob.glob('/tmp/prof/kt/**/*kernel_stats.csv', recursive=True)[0])))
out=[]
for r in rows[1:]:
    m=re.search(r'(k_[A-Za-z0-9_]+(<[^>]*>)?)', r[0])
    if not m: continue
    out.append((float(r[2])/1e6, m.group(1), int(r[1]), float(r[3])/1e6))
out.sort(reverse=True)
for t,n,c,a in out[:45]:
    print('%-40s calls %4d avg %8.3f ms total %9.1f ms'%(n,c,a,t))
PY
