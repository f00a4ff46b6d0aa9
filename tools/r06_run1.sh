set -x
mkdir -p gpurun_out/r06
python -c "import torch; p=torch.cuda.get_device_properties(0); print([a for a in dir(p) if not a.startswith('_')]); print(getattr(p,'uuid',None))" > gpurun_out/r06/props.txt 2>&1
nproc; free -g | head -2
timeout 1500 python -m pytest tests/test_gpu_config1_full.py tests/test_gpu_multi.py -x -q -s -k "config1_full or default_bench_line or (several_ranks and 2-kslam) or (starts_its_own and 4-kslam) or (starts_its_own and 2-torch)" > gpurun_out/r06/t1.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r06/t1.log
( time python bench.py ) > gpurun_out/r06/bench_a.json 2> gpurun_out/r06/bench_a.err; echo "bench rc=$?"
tail -3 gpurun_out/r06/bench_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_a.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','scaling_curve_origin')})
print(json.dumps(d['cpu_baseline'],indent=0)[:2500])
print(json.dumps(d.get('strong_n1'),indent=0)[:1500])
print(d['hot_path']['phases_ms'])
PY
