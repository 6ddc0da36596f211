for v in NOLOAD NOSTAGE NOMEM; do
  echo "== $v"
  UVIC_GPU_LIB=$PWD/uvic2.9_amd/csrc/libuvic_gpu_$v.so python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v for k,v in d['roofline']['kernel_ms'].items() if k.startswith('col')})"
done
