# option sets other than C: GPU tests of the sets + bench lines (ms per step, MOBI kernel time)
timeout -k 10 600 python -m pytest tests/test_mobi_sets.py -m gpu -x -q > gpurun_out/sets_tests.log 2>&1; tail -3 gpurun_out/sets_tests.log
for c in s37 f18; do python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-overlay --cfg $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; python3 -c "
import json;d=json.loads(open(\"gpurun_out/bench_$c.json\").read().strip().splitlines()[-1]);print(\"$c ms/step %.4f value %.3g seg4 %s\"%(d[\"ms_per_step\"],d[\"value\"],d.get(\"segment4_ms_per_step\")));print({k:round(v*1e3) for k,v in d[\"roofline\"][\"kernel_ms\"].items()})"; done
