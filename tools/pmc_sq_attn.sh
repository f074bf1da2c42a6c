cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_attn1 -o p -- ./tools/attn_bwd_bench pmc > gpurun_out/pmc_attn1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_attn2 -o p -- ./tools/attn_bwd_bench pmc > gpurun_out/pmc_attn2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for d in ('pmc_attn1','pmc_attn2'):
    f = glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:90]
        acc.setdefault(k, collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, c in acc.items():
        if 'attn' not in k: continue
        print(k)
        print('   ' + '  '.join(f"{n}={sum(v)/len(v):.4g}" for n, v in c.items()))
PY
