#!/bin/bash
# round 6: parity beyond the suite at the round's final kernel state
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/keep; mkdir -p $O
( timeout 1500 python tools/soak.py 700 71000 2>&1 | tail -2 ) > $O/r06z_soak_hot_path.txt; cat $O/r06z_soak_hot_path.txt
( KSLAM_JOIN=merge timeout 600 python tools/soak.py 200 72000 2>&1 | tail -2 ) > $O/r06z_soak_hot_path_merge_join.txt; cat $O/r06z_soak_hot_path_merge_join.txt
( timeout 700 python tools/soak_tail.py 300 2>&1 | tail -3 ) > $O/r06z_soak_tail.txt; cat $O/r06z_soak_tail.txt
( timeout 700 python tools/soak_e2e.py 200 2>&1 | tail -3 ) > $O/r06z_soak_e2e.txt; cat $O/r06z_soak_e2e.txt
( timeout 500 python tools/soak_samtext.py 150 2>&1 | tail -3 ) > $O/r06z_soak_samtext.txt; cat $O/r06z_soak_samtext.txt
