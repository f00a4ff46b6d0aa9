for v in base HSA_ENABLE_INTERRUPT=0 base HSA_ENABLE_INTERRUPT=0; do echo "== $v"; if [ "$v" = base ]; then python bench.py --no-cpu-baseline --no-abi-path --no-e2e --steps 8 --warmup 2 2>/dev/null; else env $v python bench.py --no-cpu-baseline --no-abi-path --no-e2e --steps 8 --warmup 2 2>/dev/null; fi | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['hot_path']; print(h['ms_per_step'], h['phases_ms'])"; done
