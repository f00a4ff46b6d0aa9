# usage: bash tools/pmc.sh "<counters>" <kernel substring>   (bench 1 step; counters only, no trace domains)
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof3; mkdir -p /tmp/prof3
rocprofv3 --pmc $1 --output-format csv -d /tmp/prof3 -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/o1 2> /tmp/e1
python3 - "$2" <<'PY'
import csv,glob,sys,collections
pat=sys.argv[1]
agg=collections.OrderedDict()
for f in glob.glob('/tmp/prof3/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            name=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('kslam::','').replace('void ','').split('(')[0]
            key=(r['Dispatch_Id'],name,r.get('Grid_Size','?'))
            agg.setdefault(key,{})[r['Counter_Name']]=float(r['Counter_Value'])
for k,v in list(agg.items())[:10]:
    print(k,{a:('%.5g'%b) for a,b in v.items()})
PY
