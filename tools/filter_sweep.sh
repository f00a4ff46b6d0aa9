# membership-filter sweep on the bench workload: size of the filter x key bytes sorted per batch
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$REPO" || exit 1; mkdir -p gpurun_out
for cfg in "0 3" "32 3" "32 2" "32 1" "32 0" "31 2" "33 2" "30 2"; do
  set -- $cfg
  KSLAM_FILTER_BITS=$1 KSLAM_SORT_BYTES=$2 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline > /tmp/fs.json 2>/tmp/fs.err
  python3 - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open('/tmp/fs.json'))
    print("filter_bits", sys.argv[1], "sort_bytes", sys.argv[2], "ms/step", j["hot_path"]["ms_per_step"], j["hot_path"]["phases_ms"], "kept", j["hot_path"]["counts"]["read_kmers_kept_by_filter"], "raw", j["hot_path"]["counts"]["overlaps_raw"], "ok", j["hot_path"]["verified"]["ok"], "index_s", j["setup_s"]["index_build"])
except Exception as e:
    print("failed", sys.argv[1:], e, open('/tmp/fs.err').read()[-500:])
PY
done
