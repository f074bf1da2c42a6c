#!/bin/bash
# rocprofv3 kernel stats of the embed-only run (BASELINE.json configs[3] shape): bash tools/prof_infer.sh [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_infer -o r -- python3 bench.py --mode infer --speakers 256 --utts 1 --samples 5 --frames 64 --steps 50 --warmup 10 > gpurun_out/prof_infer.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_infer/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows if 'ge2e' in r['Name'])
print('sum of kernel time per step (us):', tot/60/1e3)
for r in rows[:16]: print(f"  {r['Name'][:80]:80s} {int(r['Calls'])/60:4.1f}/step {float(r['AverageNs'])/1e3:8.1f} us")
PY
tail -1 gpurun_out/prof_infer.log | cut -c1-300
