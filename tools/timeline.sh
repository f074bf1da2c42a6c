#!/bin/bash
# Kernel timeline of one training step (rocprofv3 kernel trace): bash tools/timeline.sh <tag> [ENV=VALUE ...]  -> gpurun_out/timeline_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -o r -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/tl_$tag.log 2>&1 || exit 1
python3 tools/timeline.py $(ls gpurun_out/tl_$tag/*kernel_trace.csv | head -1) > gpurun_out/timeline_$tag.txt
tail -1 gpurun_out/timeline_$tag.txt
