# per-dispatch durations of selected kernels (rocprofv3 kernel trace), compact output
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof2; mkdir -p /tmp/prof2
KSLAM_DEBUG=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof2 -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/o1 2> /tmp/e1
grep kslam /tmp/e1
python3 - "$1" <<'PY'
import csv,glob,sys
pat=sys.argv[1]
f=glob.glob('/tmp/prof2/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows:
    n=r['Kernel_Name']
    if 'kslam' in n and any(p in n for p in pat.split(',')):
        nm=n.replace('(anonymous namespace)::','').replace('kslam::','').replace('void ','').split('(')[0]
        print(nm.ljust(28),'grid',r.get('Grid_Size','?').rjust(10),'lds',r.get('LDS_Block_Size','?').rjust(7),'vgpr',r.get('VGPR_Count','?'),'us=%9.1f'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
