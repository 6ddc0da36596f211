#!/bin/bash
# All measurement artefacts of a round from ONE bench command, written to gpurun_out/prof/ :
#   bench.json            the bench line (with cpu_baseline)
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of the same command
#   clinic_kernel_stats.csv  the same for tools/clinic_loop.py (state + clinic)
#   pmc_*                 separate --pmc passes (SQ, FETCH_SIZE, WRITE_SIZE, TCC) + calibration passes
#   traffic.json, pmc_summary.txt   from tools/pmc_summary.py
# Copy what is to be judged into profiles/ (gpurun_out/ is scratch).
R=$PWD; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
BENCH="python3 $R/bench.py --steps 64 --warmup 4"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH --no-cpu-baseline --no-overlay > $O/stats.log 2>&1 || exit 1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
# the momentum row (state + clinic, polar filter and sbc accumulation on): its own loop, same counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_clinic -- python3 $R/tools/clinic_loop.py 200 > $O/stats_clinic.log 2>&1 || exit 1
cp $(ls $O/stats_clinic/*/*kernel_stats.csv | head -1) $O/clinic_kernel_stats.csv
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlay > $O/pmc_$name.log 2>&1 || exit 1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
runc() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -- python3 $R/tools/clinic_loop.py 3 > $O/pmc_clinic_$name.log 2>&1 || exit 1; }
runc fetch FETCH_SIZE
runc write WRITE_SIZE
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc/calib_fetch -- $R/tools/calib_traffic.bin 64 > $O/pmc_calib_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc/calib_write -- $R/tools/calib_traffic.bin 64 > $O/pmc_calib_write.log 2>&1 || exit 1
cd $R
python3 tools/pmc_summary.py $O/pmc $O/traffic.json > $O/pmc_summary.txt 2>&1 || exit 1
cp $O/traffic.json $R/gpurun_out/traffic_latest.json
UVIC_TRAFFIC_JSON=$O/traffic.json python3 bench.py --steps 64 --warmup 4 > $O/bench.json 2> $O/bench.err || exit 1
tail -1 $O/bench.json; tail -25 $O/pmc_summary.txt
