#!/bin/bash
# PMC counters of one kernel under the sam_pipeline leg:  bash tools/pmc_kernel.sh KERNEL "COUNTER COUNTER ..."
K=${1:-k_row_details}; C=${2:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS}
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1; rm -rf /tmp/prof_pk
rocprofv3 --pmc $C --output-format csv -d /tmp/prof_pk -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-abi-path --no-full-pipeline > /tmp/o_pk 2> /tmp/e_pk
python3 - "$K" <<'PY'
import csv, glob, sys
K = sys.argv[1]
agg = {}
for f in glob.glob('/tmp/prof_pk/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r['Kernel_Name']:
            a = agg.setdefault(r['Counter_Name'], [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in sorted(agg.items()):
    print('%-24s dispatches %4d mean %.6g' % (k, n, v / n))
PY
