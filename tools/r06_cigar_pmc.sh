#!/bin/bash
# round 6: what holds the narrow-band CIGAR kernels -- SQ wave-cycle breakdown per kernel (two --pmc passes, counters only)
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$REPO" || exit 1
O=gpurun_out/r06; mkdir -p $O; rm -f $O/cigar_pmc_raw.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/kp
  rocprofv3 --pmc $set --output-format csv -d /tmp/kp -o x -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-pipeline > /tmp/kp.json 2> /tmp/kp.err
  python3 - <<'PY' >> gpurun_out/r06/cigar_pmc_raw.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('/tmp/kp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if any(k in n for k in ('k_banded_lds', 'k_cigar_systolic', 'k_systolic_traceback', 'k_sw_band', 'k_join_fill', 'k_extract_filter')):
            short = n.replace('(anonymous namespace)::', '').replace('kslam_api::', '').replace('kslam::', '').replace('void ', '').split('(')[0]
            a = agg[(short, r['Counter_Name'])]; a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, v) in sorted(agg.items()):
    print(k, c, n, v / n)
PY
done
cat $O/cigar_pmc_raw.txt | grep "k_banded_lds"
