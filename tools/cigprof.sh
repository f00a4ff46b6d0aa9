REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; mkdir -p gpurun_out
for mode in auto slab; do
  rm -rf /tmp/prof_$mode
  if [ $mode = slab ]; then export KSLAM_CIGAR_DIRS=slab; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$mode -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/o_$mode.json 2>/tmp/e_$mode
  python3 - $mode <<'PY'
import csv, glob, sys, json
mode = sys.argv[1]
f = glob.glob('/tmp/prof_%s/**/*kernel_trace.csv' % mode, recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_banded_lds' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
print(mode, "k_banded_lds dispatches (ms):", [round(x, 2) for x in d[-6:]], "lds:", [r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', '?')) for r in rows[-6:]])
j = json.load(open('/tmp/o_%s.json' % mode)); print(mode, j['hot_path']['ms_per_step'], j['hot_path']['phases_ms'])
PY
done
