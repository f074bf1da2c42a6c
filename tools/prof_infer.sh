cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for wv in 4 8; do
  export GE2E_FFN_WV=$wv
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_infer_wv$wv -o r -- python3 bench.py --mode infer --steps 20 --warmup 5 > gpurun_out/prof_infer_wv$wv.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv,glob
for wv in (4,8):
    f=glob.glob(f'gpurun_out/prof_infer_wv{wv}/**/*kernel_stats.csv',recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    print('wv',wv)
    for r in rows[:8]: print(f"  {r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
