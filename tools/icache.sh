# instruction-cache counters of the time loop (per kernel, mean per launch) -> gpurun_out/icache/
R=$PWD; O=$R/gpurun_out/icache; rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if k.startswith("k_"):
        m = {c: sum(v) / len(v) for c, v in agg[k].items()}
        req = m.get("SQC_ICACHE_REQ", 0) or 1
        print(f"{k:20s} req={m.get('SQC_ICACHE_REQ',0):.3g} hit={m.get('SQC_ICACHE_HITS',0):.3g} miss={m.get('SQC_ICACHE_MISSES',0):.3g} dup={m.get('SQC_ICACHE_MISSES_DUPLICATE',0):.3g} missrate={(m.get('SQC_ICACHE_MISSES',0)+m.get('SQC_ICACHE_MISSES_DUPLICATE',0))/req:.3f} wait_inst/wave_cycles={m.get('SQ_WAIT_INST_ANY',0)/(m.get('SQ_WAVE_CYCLES',1) or 1):.3f}")
PY
