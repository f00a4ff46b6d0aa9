#!/bin/bash
# round 6: parity beyond the suite at the round's final kernel state (usage: bash tools/r06_soaks.sh [hot-path rounds] [first seed])
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/keep; mkdir -p $O
N=${1:-700}; S=${2:-71000}
( timeout 3000 python tools/soak.py $N $S 2>&1 | tail -2 ) > $O/r06z_soak_hot_path.txt; cat $O/r06z_soak_hot_path.txt
( KSLAM_JOIN=merge timeout 600 python tools/soak.py 150 72000 2>&1 | tail -2 ) > $O/r06z_soak_hot_path_merge_join.txt; cat $O/r06z_soak_hot_path_merge_join.txt
( KSLAM_SWEEP_ROOM=0 timeout 600 python tools/soak.py 150 73000 2>&1 | tail -2 ) > $O/r06z_soak_hot_path_sweep_room0.txt; cat $O/r06z_soak_hot_path_sweep_room0.txt
( timeout 500 python tools/soak_e2e.py 150 2>&1 | tail -3 ) > $O/r06z_soak_e2e.txt; cat $O/r06z_soak_e2e.txt
