set -e
# Round-end measurement set (run on the MI355X through gpurun): bench lines, rocprofv3 kernel stats, two PMC passes for HBM traffic.
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo default done
python bench.py --no-cpu-baseline --precision fp32 --steps 10 --warmup 3 > $O/bench_fp32.json 2>/dev/null
python bench.py --no-cpu-baseline --precision fp32x3 --steps 10 --warmup 3 > $O/bench_fp32x3.json 2>/dev/null
python bench.py --no-cpu-baseline --precision fp16 > $O/bench_fp16.json 2>/dev/null
python bench.py --no-cpu-baseline --speakers 256 --utts 10 --frames 180 --steps 10 --warmup 3 > $O/bench_cfg5_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --precision fp16 --speakers 256 --utts 10 --frames 180 --steps 10 --warmup 3 > $O/bench_cfg5_fp16.json 2>/dev/null
python bench.py --mode infer --speakers 256 --utts 1 --samples 5 --frames 64 --steps 50 --warmup 10 > $O/bench_infer_bf16.json 2>/dev/null
python bench.py --mode infer --precision fp32 --speakers 256 --utts 1 --samples 5 --frames 64 --steps 50 --warmup 10 > $O/bench_infer_fp32.json 2>/dev/null
python bench.py --mode infer --precision fp32x3 --speakers 256 --utts 1 --samples 5 --frames 64 --steps 50 --warmup 10 > $O/bench_infer_fp32x3.json 2>/dev/null
for k in gemm gemm_ln ffn attn_fwd attn_bwd; do python bench.py --no-cpu-baseline --roofline-kernel $k > $O/bench_roof_$k.json 2>/dev/null; done
echo benches done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $R -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_stats.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > $O/pmc_f.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > $O/pmc_w.log 2>&1
echo write done
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/$R
cat $O/bench_*.json > $O/${R}_bench_lines.jsonl
ls $O
