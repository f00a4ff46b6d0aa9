#!/bin/bash
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "merge_join or filter_built or every_kernel_variant or group_route or overlap_keys" > $O/t3.log 2>&1; echo "pytest rc=$?"; tail -5 $O/t3.log
BA="--steps 3 --warmup 1 --no-cpu-baseline --no-full-pipeline"
show() { python3 - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], j['hot_path']['phases_ms'], j['roofline']['index_sort']['index_build_ms'], j['roofline']['index_sort']['frac'], j['hot_path']['verified']['ok'])
PY
}
stats() { python3 - "$1" "$2" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in sys.argv[2].split(',')):
        print('   ', n.split('(')[0].split('::')[-1][:40], n[n.find('<'):n.find('>')+1][:12] if '<' in n.split('(')[0] else '', r['Calls'], round(float(r['AverageNs'])/1e6, 4), 'ms avg', round(float(r['TotalDurationNs'])/1e6, 3), 'ms total')
PY
}
# (a) index build: new (blocks + 4-tile byte histograms) vs the atomic filter build
for mode in blocks atomics; do
  rm -rf /tmp/kp
  if [ $mode = atomics ]; then export KSLAM_FILTER_BUILD=atomics; else unset KSLAM_FILTER_BUILD; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o x -- python3 bench.py $BA > $O/index_${mode}.json 2> /tmp/kp.err
  cp $(find /tmp/kp -name '*kernel_stats.csv' | head -1) $O/index_${mode}_kernel_stats.csv
  show $O/index_${mode}.json
  stats $O/index_${mode}_kernel_stats.csv "k_filter,k_tile_hist_bytes_setup,k_tile_hist_setup,k_scatter_setup,k_split,k_bucket,k_join"
done
unset KSLAM_FILTER_BUILD
# (b) the probe before / after this round's refactoring of its emission, same box
cp k-slam_amd/libkslam_hip.so /tmp/new.so
for lib in new old new old; do
  if [ $lib = old ]; then cp k-slam_amd/libkslam_hip_oldjoin.so k-slam_amd/libkslam_hip.so; else cp /tmp/new.so k-slam_amd/libkslam_hip.so; fi
  rm -rf /tmp/kp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o x -- python3 bench.py $BA > /tmp/j.json 2> /tmp/kp.err
  echo "join lib=$lib"; stats $(find /tmp/kp -name '*kernel_stats.csv' | head -1) "k_join"
done
cp /tmp/new.so k-slam_amd/libkslam_hip.so
# (c) where the GPU waits for the host inside one alignment call
bash tools/gaps.sh > $O/gaps.txt 2>&1; head -40 $O/gaps.txt
