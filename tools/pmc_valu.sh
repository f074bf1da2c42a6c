#!/bin/bash
# Per-kernel vector-ALU activity of the training step (kernels alone on the chip: GE2E_NO_OVERLAP=1):  bash tools/pmc_valu.sh <tag>
#   valu_share = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): issue cycles of vector instructions against the SIMD cycles of the launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-valu}
export GE2E_DEV_SWITCHES=1 GE2E_NO_OVERLAP=1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o v -- python3 bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_$tag/**/*counter_collection.csv', recursive=True)[0]
acc = collections.OrderedDict(); dur = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'ge2e' not in k: continue
    acc.setdefault(k, collections.defaultdict(float))[r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id'])); dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        acc[k]['n'] += 1
rows = []
for k, c in acc.items():
    n = c['n']; simd_cyc = c['GRBM_GUI_ACTIVE'] / 8 * 1024
    rows.append((dur[k] / 3, k[:70], n / 3, dur[k] / n, 4 * c['SQ_ACTIVE_INST_VALU'] / simd_cyc, c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cyc, c['SQ_ACTIVE_INST_VALU'] / max(c['SQ_WAVE_CYCLES'], 1), c['SQ_WAIT_ANY'] / max(c['SQ_WAVE_CYCLES'], 1), c['SQ_WAIT_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1), c['SQ_WAVE_CYCLES'] / max(c['SQ_BUSY_CYCLES'], 1)))
rows.sort(reverse=True)
print(f"{'kernel':70s} {'n/step':>6s} {'us':>8s} {'valu':>6s} {'mfma':>6s} {'w.valu':>6s} {'w.wait':>6s} {'w.stall':>7s} {'wv/bz':>6s}")
for r in rows[:32]: print(f"{r[1]:70s} {r[2]:6.1f} {r[3]:8.1f} {r[4]:6.2f} {r[5]:6.2f} {r[6]:6.2f} {r[7]:6.2f} {r[8]:7.2f} {r[9]:6.1f}")
PY
