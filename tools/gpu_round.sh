#!/bin/bash
# tests + bench (+ variants given as "NAME:ENV=VAL,ENV=VAL" arguments) + kernel timeline, into gpurun_out/$1/
# usage on the GPU box:  bash tools/gpu_round.sh r2x [notests] [notl] [var1:UVIC_X=1 ...]
O=gpurun_out/$1; shift; mkdir -p $O
TESTS=1; TL=1
while [ "$1" = "notests" ] || [ "$1" = "notl" ]; do [ "$1" = "notests" ] && TESTS=0; [ "$1" = "notl" ] && TL=0; shift; done
if [ $TESTS = 1 ]; then python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log; fi
run() { name=$1; envs=$2; ( [ -n "$envs" ] && export ${envs//,/ }; python bench.py --steps 64 --warmup 4 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err ) || { echo "bench $name failed"; tail -5 $O/bench_$name.err; return; }
  python3 - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
print("$name", "ms/step %.4f submit %.4f" % (d["ms_per_step"], d["host_submit_ms_per_step"]))
print("  loop ", {k: round(v*1e3) for k, v in d["roofline"]["kernel_ms"].items()})
print("  alone", {k: round(v*1e3) for k, v in d["roofline"]["kernel_ms_isolated"].items()})
PY
}
run base ""
for v in "$@"; do run "${v%%:*}" "${v#*:}"; done
if [ $TL = 1 ]; then bash tools/timeline.sh > $O/tl.log 2>&1; cp gpurun_out/tl/timeline.txt $O/timeline.txt; tail -48 $O/timeline.txt; fi
