#!/bin/bash
# Where do the chained FFN forward's fetches come from?  L2 hit / miss and fabric read requests per launch of the two shipped 4-wave kernels:
#   bash tools/pmc_tcc_ffn.sh        (rocprofv3 --pmc with --kernel-trace only; FETCH_SIZE in its own pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcc_ffn -o t -- ./tools/ffn_bench 153600 pmc > gpurun_out/pmc_tcc_ffn.log 2>&1 || { tail -5 gpurun_out/pmc_tcc_ffn.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_tcc_ffn_f -o f -- ./tools/ffn_bench 153600 pmc > gpurun_out/pmc_tcc_ffn_f.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
acc = collections.OrderedDict()
for d in ('pmc_tcc_ffn', 'pmc_tcc_ffn_f'):
    f = glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if 'ffn_chain' not in r['Kernel_Name']: continue
        acc.setdefault(r['Kernel_Name'][:70], collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k)
    print("   per launch: L2 requests %.3g, hits %.3g, misses %.3g (hit rate %.3f); fabric read requests %.3g (x 64 B = %.1f MB, x 128 B = %.1f MB); FETCH_SIZE %.1f MB raw, %.1f MB doubled"
          % (m.get('TCC_REQ_sum', 0), m.get('TCC_HIT_sum', 0), m.get('TCC_MISS_sum', 0), m.get('TCC_HIT_sum', 0) / max(1.0, m.get('TCC_HIT_sum', 0) + m.get('TCC_MISS_sum', 0)),
             m.get('TCC_EA0_RDREQ_sum', 0), m.get('TCC_EA0_RDREQ_sum', 0) * 64 / 1e6, m.get('TCC_EA0_RDREQ_sum', 0) * 128 / 1e6, m.get('FETCH_SIZE', 0) * 1024 / 1e6, m.get('FETCH_SIZE', 0) * 2048 / 1e6))
PY
