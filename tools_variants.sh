# bench the MOBI-team register budgets x column-kernel variants (diagnostic)
for w in 0 2 3; do for pf in 0 1; do
  echo "team_waves_per_eu=$w col_prefetch=$pf"
  UVIC_GPU_LIB=$PWD/uvic2.9_amd/csrc/libuvic_gpu_w$w.so UVIC_COL_PREFETCH=$pf python bench.py --steps 64 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  ms/step', round(d['ms_per_step'],4), ' live colfct', r['kernel_ms']['colfct'], 'colupd', r['kernel_ms']['colupd'], 'mobi', r['kernel_ms']['mobi'], ' iso colfct', r['kernel_ms_isolated']['colfct'], 'mobi', r['kernel_ms_isolated']['mobi'])"
done; done
