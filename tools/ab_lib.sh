#!/bin/bash
# A/B of two BUILDS on one box: bash tools/ab_lib.sh a b   (variants from tools/build_variant.py under tools/abl/<name>/).  A variant is loaded
# through GE2E_LIB_OVERRIDE (honoured only with GE2E_DEV_SWITCHES=1; the loader prints which binary it took): the in-tree library is never touched.
set -e
export GE2E_DEV_SWITCHES=1
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
for v in "$@"; do
  GE2E_LIB_OVERRIDE=$PWD/tools/abl/$v/libge2e_hip.so timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline ${AB_ARGS} > gpurun_out/ab/lib_${v}_$rep.json 2> gpurun_out/ab/lib_${v}_$rep.err || { tail -5 gpurun_out/ab/lib_${v}_$rep.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/lib_${v}_$rep.json").read().strip().splitlines()[-1])
print("$v", $rep, d["ms_per_step"], d["value"])
PY
done; done
