mkdir -p gpurun_out/slab
for n in 2 4 8; do python bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-overlay --one-slab-of $n > gpurun_out/slab/s$n.json 2> gpurun_out/slab/s$n.err && python3 -c "
import json;d=json.loads(open('gpurun_out/slab/s$n.json').read().strip().splitlines()[-1]);print('slab 1 of $n: ms/step %.4f'%d['ms_per_step'], {k:round(v*1e3) for k,v in d['roofline']['kernel_ms_isolated'].items()})"; done
python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-overlay --cfg c30 --grid 202x202x32 > gpurun_out/slab/big.json 2> gpurun_out/slab/big.err && python3 -c "
import json;d=json.loads(open('gpurun_out/slab/big.json').read().strip().splitlines()[-1]);print('202x202x32: ms/step %.4f value %.3g'%(d['ms_per_step'],d['value']), {k:round(v*1e3) for k,v in d['roofline']['kernel_ms_isolated'].items()})"
python bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-overlay --cfg perf15 > gpurun_out/slab/p15.json 2> gpurun_out/slab/p15.err && python3 -c "
import json;d=json.loads(open('gpurun_out/slab/p15.json').read().strip().splitlines()[-1]);print('perf15: ms/step %.4f value %.3g'%(d['ms_per_step'],d['value']))"
python bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-overlay --cfg p2 > gpurun_out/slab/p2.json 2> gpurun_out/slab/p2.err && python3 -c "
import json;d=json.loads(open('gpurun_out/slab/p2.json').read().strip().splitlines()[-1]);print('p2: ms/step %.4f value %.3g'%(d['ms_per_step'],d['value']))"
