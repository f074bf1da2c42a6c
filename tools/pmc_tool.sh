#!/bin/bash
# HBM bytes per kernel of a standalone tool: bash tools/pmc_tool.sh <tag> <binary> [args]   (two PMC passes, FETCH doubled per the gfx950 note)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmct_${tag}_f -o f -- "$@" > gpurun_out/pmct_${tag}_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmct_${tag}_w -o w -- "$@" > gpurun_out/pmct_${tag}_w.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
def load(d, name):
    f = glob.glob(f'gpurun_out/pmct_${tag}_{d}/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != name: continue
        k = (r['Kernel_Name'][:60], r['Grid_Size'])
        acc.setdefault(k, []).append(float(r['Counter_Value']))
    return acc
F, W = load('f', 'FETCH_SIZE'), load('w', 'WRITE_SIZE')
for k in F:
    f = sum(F[k]) / len(F[k]) * 1024 * 2 / 1e6       # KB units, doubled on gfx950 for wide streaming reads
    w = sum(W.get(k, [0])) / max(1, len(W.get(k, [0]))) * 1024 / 1e6
    print(f"{k[0]:60s} grid {k[1]:>8s} n={len(F[k]):3d}  fetch {f:8.1f} MB  write {w:8.1f} MB")
PY
