for cfg in "1 1024 512" "1 512 512" "1 512 256" "1 256 256" "2 512 256"; do
  set -- $cfg
  echo "== nchunk=$1 fct_threads=$2 upd_threads=$3"
  UVIC_NCHUNK=$1 UVIC_FCT_THREADS=$2 UVIC_UPD_THREADS=$3 python bench.py --steps 48 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done
