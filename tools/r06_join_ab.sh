#!/bin/bash
# round 6: the probe (k_join_fill) against the merge (k_join_merge, KSLAM_JOIN=merge): parity, kernel times, HBM counters
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
BA="--steps 3 --warmup 1 --no-cpu-baseline --no-full-pipeline"
KSLAM_JOIN=merge timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q > $O/join_merge_parity.log 2>&1; echo "merge parity rc=$?"; tail -3 $O/join_merge_parity.log
for mode in probe merge; do
  for wl in c1 repeats c2; do
    case $wl in c1) A="";; repeats) A="--repeats";; c2) A="--config 2";; esac
    rm -rf /tmp/kp
    if [ $mode = merge ]; then export KSLAM_JOIN=merge; else unset KSLAM_JOIN; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o x -- python3 bench.py $BA $A > $O/join_${mode}_${wl}.json 2> /tmp/kp.err
    cp $(find /tmp/kp -name '*kernel_stats.csv' | head -1) $O/join_${mode}_${wl}_kernel_stats.csv
  done
  for ctr in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    rm -rf /tmp/kp
    rocprofv3 --pmc $ctr --output-format csv -d /tmp/kp -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-pipeline > /tmp/kp.json 2> /tmp/kp.err
    python3 - "$mode" "$ctr" <<'PY' >> $O/join_pmc.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('/tmp/kp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_join' in r['Kernel_Name']:
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
            a = agg[(name, r['Counter_Name'])]; a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, v) in sorted(agg.items()):
    print(sys.argv[1], k, c, 'dispatches', n, 'mean', v / n)
PY
  done
done
unset KSLAM_JOIN
cat $O/join_pmc.txt
python3 - <<'PY'
import csv, json, glob
for mode in ('probe', 'merge'):
    for wl in ('c1', 'repeats', 'c2'):
        try:
            rows = list(csv.DictReader(open('gpurun_out/r06/join_%s_%s_kernel_stats.csv' % (mode, wl))))
            j = json.loads(open('gpurun_out/r06/join_%s_%s.json' % (mode, wl)).read().strip().splitlines()[-1])
        except Exception as e:
            print(mode, wl, 'missing', e); continue
        k = [r for r in rows if 'k_join' in r['Name']]
        print(mode, wl, [(r['Name'].replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0], r['Calls'], round(float(r['AverageNs']) / 1e6, 4)) for r in k],
              j['hot_path']['phases_ms']['ms_join'], j['hot_path']['ms_per_step'], j['hot_path']['verified']['ok'], j['hot_path']['counts']['overlaps_raw'])
PY
