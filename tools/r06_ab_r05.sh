#!/bin/bash
# round 6 against round 5 on ONE box: the hot path (resident batch) with each round's library, alternating.
# The round-5 library is not in the tree; build it first (in the container, the .so travels with the snapshot):
#   mkdir -p /tmp/r05src && git archive c1833de k-slam_amd include | tar -x -C /tmp/r05src && make -C /tmp/r05src/k-slam_amd/csrc -j8 ../libkslam_hip.so
#   cp /tmp/r05src/k-slam_amd/libkslam_hip.so k-slam_amd/libkslam_hip_r05.so      (git-ignored; same C ABI, loaded through KSLAM_LIB)
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O
BA="--steps 10 --warmup 3 --no-cpu-baseline --no-full-pipeline"
for i in 1 2 3; do
  for lib in r05 r06; do
    if [ $lib = r05 ]; then export KSLAM_LIB=$REPO/k-slam_amd/libkslam_hip_r05.so; else unset KSLAM_LIB; fi
    python bench.py $BA $1 > $O/ab_$lib.json 2>/dev/null
    python3 -c "
import json; j=json.loads(open('$O/ab_$lib.json').read().strip().splitlines()[-1]); i=j['roofline']['index_sort']
print('$lib', j['hot_path']['phases_ms'], 'index', i['index_build_ms'], 'frac', i['frac'], j['hot_path']['verified']['ok'])"
  done
done
unset KSLAM_LIB
