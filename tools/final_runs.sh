# the round's closing measurements, one call: bash tools/final_runs.sh rNNz
R=${1:-r05z}
mkdir -p gpurun_out/keep
python bench.py --steps 20 --warmup 5 > gpurun_out/keep/${R}_bench.json 2> gpurun_out/${R}_bench.err; tail -c 400 gpurun_out/keep/${R}_bench.json
python bench.py --config 2 --no-cpu-baseline > gpurun_out/keep/${R}_bench_config2.json 2> gpurun_out/${R}_c2.err
python bench.py --config 4 --no-cpu-baseline > gpurun_out/keep/${R}_bench_config4.json 2> gpurun_out/${R}_c4.err
python bench.py --strong --no-cpu-baseline > gpurun_out/keep/${R}_bench_strong.json 2> gpurun_out/${R}_st.err
KSLAM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --strong --no-cpu-baseline > gpurun_out/keep/${R}_bench_strong_rccl_world1.json 2> gpurun_out/${R}_st1.err
python bench.py --repeats --no-cpu-baseline > gpurun_out/keep/${R}_bench_repeats.json 2> gpurun_out/${R}_rp.err
python bench.py --legs all --strong-n1 off --no-cpu-baseline > gpurun_out/keep/${R}_bench_all_legs.json 2> gpurun_out/${R}_al.err
for f in bench bench_config2 bench_config4 bench_strong bench_strong_rccl_world1 bench_repeats bench_all_legs; do python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/keep/${R}_'+'$f'+'.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['hot_path']['reads_per_s'])
except Exception as e: print('$f', 'FAILED', e)"; done
