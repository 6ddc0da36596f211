# kernels and copies of the resident Fortran overlay's `tracer` calls, back to back -> gpurun_out/otl/
R=$PWD; rm -rf $R/gpurun_out/otl; mkdir -p $R/gpurun_out/otl; cd /tmp && export TMPDIR=/tmp
export UVIC_RESIDENT=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/otl/kt -- python3 $R/tools/overlay_time.py 24 > $R/gpurun_out/otl/run.log 2>&1
cd $R; python tools/overlay_tl.py gpurun_out/otl/kt > gpurun_out/otl/timeline.txt 2>&1; head -3 gpurun_out/otl/run.log; tail -3 gpurun_out/otl/run.log
