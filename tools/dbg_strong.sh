mkdir -p /tmp/keep; KSLAM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29591 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 KSLAM_BENCH_KEEP_FILES=/tmp/keep python bench.py --strong --total-pairs 40000 --species 4 --strains 3 --genome-len 300000 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/dbg.json 2> /tmp/dbg.err; tail -3 /tmp/dbg.err; ls -la /tmp/keep
python3 - <<'PY'
import glob
fs=sorted(glob.glob('/tmp/keep/*'))
print(fs)
full=[f for f in fs if f.endswith('strong.sam')][0]
part=[f for f in fs if 'part0.sam' in f and not f.endswith('_PerRead')][0]
a=open(full,'rb').read(); b=open(part,'rb').read()
print(len(a),len(b))
la=a.split(b'\n'); lb=b.split(b'\n')
n=0
for i,(x,y) in enumerate(zip(la,lb)):
    if x!=y:
        print(i); print(x[:300]); print(y[:300]); n+=1
        if n>3: break
pa=open(full+'_PerRead','rb').read(); pb=open(part+'_PerRead','rb').read()
print('perread',len(pa),len(pb),pa==pb, pa[:60], pb[:60])
PY
python3 -c "
import json;d=json.loads(open('/tmp/dbg.json').read().strip().splitlines()[-1]);print(d.get('verified_classified'));print(d.get('classified_sharded'));print(d['classified_rank0_tail'])"
