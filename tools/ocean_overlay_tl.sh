# kernels and copies of both Fortran overlays in the ocean loop (tools/ocean_overlay_time.py [cfg]) -> gpurun_out/ootl/
R=$PWD; rm -rf $R/gpurun_out/ootl; mkdir -p $R/gpurun_out/ootl; cd /tmp && export TMPDIR=/tmp
export UVIC_RESIDENT=${UVIC_RESIDENT:-2}
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/ootl/kt -- python3 $R/tools/ocean_overlay_time.py 16 ${1:-t30} > $R/gpurun_out/ootl/run.log 2>&1
cd $R; python tools/overlay_tl.py gpurun_out/ootl/kt 2 > gpurun_out/ootl/timeline.txt 2>&1; tail -2 gpurun_out/ootl/run.log
