for m in 248 240 248 240; do echo "== KSLAM_CIGAR_SYS=$m"; KSLAM_CIGAR_SYS=$m python bench.py --no-cpu-baseline --no-abi-path --no-sam-pipeline --no-full-pipeline --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['hot_path']; print(h['phases_ms'], h['verified']['ok'])"; done
