#!/bin/bash
# Where the narrow-band CIGAR kernel's time goes: k_banded_lds<2> with parts switched off (KSLAM_CIGAR_VARIANT:
# 0 full, 1 no traceback, 2 staging only, 4 header loads only, 3 launch floor).  Results are meaningless with a variant on.
# Needs the measurement-only library: make -C k-slam_amd/csrc ABLATE=1 (the product build has no such switches).
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
export KSLAM_LIB="$REPO/k-slam_amd/libkslam_hip_ablate.so"
[ -f "$KSLAM_LIB" ] || make -C k-slam_amd/csrc -s -j8 ABLATE=1 || exit 1
for v in 0 1 2 4 3; do
  rm -rf /tmp/prof_ca
  KSLAM_CIGAR_VARIANT=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ca -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/o_ca.json 2>/tmp/e_ca
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
rows = list(csv.reader(open(glob.glob('/tmp/prof_ca/**/*kernel_stats.csv', recursive=True)[0])))
for r in rows[1:]:
    if 'k_banded_lds<2>' in r[0] or 'k_cigar_systolic<160, 8, 4' in r[0] or 'k_cigar_systolic<160, 8, 8' in r[0]:
        print('variant', v, r[0].split('(')[0][-34:], 'calls', r[1], 'avg_us %.1f max_us %.1f' % (float(r[3]) / 1e3, float(r[6]) / 1e3))
PY
done
