#!/bin/bash
# A/B of two BUILDS on one box: bash tools/ab_lib.sh a b   (variants from tools/build_variant.py: tools/abl/<name>/{libge2e_hip.so,csrc/*} are
# copied into place per run, so the loader's source-hash check holds; the GPU box's copy of the tree is left with the LAST variant)
set -e
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
for v in "$@"; do
  cp tools/abl/$v/libge2e_hip.so speaker_embedding_torch_amd/libge2e_hip.so; cp tools/abl/$v/csrc/* speaker_embedding_torch_amd/csrc/
  timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline ${AB_ARGS} > gpurun_out/ab/lib_${v}_$rep.json 2> gpurun_out/ab/lib_${v}_$rep.err || { tail -5 gpurun_out/ab/lib_${v}_$rep.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/lib_${v}_$rep.json").read().strip().splitlines()[-1])
print("$v", $rep, d["ms_per_step"], d["value"])
PY
done; done
