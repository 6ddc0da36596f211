# kernel timeline of a few bench steps (start/end per dispatch) -> gpurun_out/tl/
R=$PWD; rm -rf $R/gpurun_out/tl; mkdir -p $R/gpurun_out/tl; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/kt -- python3 $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-overlay > $R/gpurun_out/tl/kt.log 2>&1
cd $R; python tools/timeline.py gpurun_out/tl/kt > gpurun_out/tl/timeline.txt 2>&1; tail -60 gpurun_out/tl/timeline.txt
