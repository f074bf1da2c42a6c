#!/bin/bash
# rocprofv3 kernel stats of the training step under an environment switch: bash tools/prof_step.sh <tag> [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o r -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline ${BENCH_ARGS} > gpurun_out/prof_$tag.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,json
f=glob.glob('gpurun_out/prof_$tag/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    if 'ge2e' in r['Name'] or 'rocclr' in r['Name']: tot+=float(r['TotalDurationNs'])
print('$tag', 'sum of kernel time per step (us):', tot/25/1e3)
for r in rows[:24]: print(f"  {r['Name'][:80]:80s} {int(r['Calls'])//25:>3d}/step {float(r['AverageNs'])/1e3:9.1f} us")
print(open('gpurun_out/prof_$tag.log').read().strip().splitlines()[-1][:200])
PY
