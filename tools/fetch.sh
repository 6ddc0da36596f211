# FETCH_SIZE / WRITE_SIZE per launch of the column kernels (raw KB; fetch counts 1/2 for 8 B/lane)
R=$PWD; O=$R/gpurun_out/fetch; rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/w.log 2>&1
cd $R; python3 - <<'PY'
import csv, glob, collections
for d, cn in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/fetch/{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cn: agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        if k.startswith("k_col") or k.startswith("k_coef"): print(f"{cn} {k:12s} {sum(v)/len(v)*1024*(2 if d=='f' else 1)/1e6:8.1f} MB per launch (corrected)")
PY
