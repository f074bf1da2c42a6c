#!/bin/bash
# A/B of bench.py under different library options on ONE box (run-to-run noise between boxes is ~1 %):
#   bash tools/ab.sh "base:" "wv8:GE2E_FFN_WV=8" ...        (name:ENV=VALUE[ ENV=VALUE]); two interleaved repetitions
# GE2E_DEV_SWITCHES=1 makes the ctypes loader forward GE2E_<NAME>=value to ge2e_set_option (the library itself never reads the environment).
set -e
export GE2E_DEV_SWITCHES=1
mkdir -p gpurun_out/ab
[ $# -gt 0 ] || set -- "base:"
for rep in 1 2; do
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline ${AB_ARGS} > gpurun_out/ab/bench_${name}_$rep.json 2> gpurun_out/ab/bench_${name}_$rep.err || { tail -5 gpurun_out/ab/bench_${name}_$rep.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/bench_${name}_$rep.json").read().strip().splitlines()[-1])
print("$name", $rep, d["ms_per_step"], d["value"])
PY
done; done
