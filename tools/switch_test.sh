# the train-step parity tests once per library option (profiles/rNN_switch_test.txt)
set -e
export GE2E_DEV_SWITCHES=1
for sw in "GE2E_NO_SK_GEMM=1" "GE2E_NO_COLSUM_END=1" "GE2E_NO_FFN_CHAIN_BWD=1" "GE2E_NO_PRENET_FUSE=1" "GE2E_NO_OVERLAP=1" "GE2E_NO_FFN_CHAIN=1" "GE2E_NO_REDUCE_BATCH=1" "GE2E_NO_WGRAD_KS=1" "GE2E_NO_MASKBITS=1" "GE2E_NO_LNFUSE=1" "GE2E_NO_EVENT_BIND=1" "GE2E_FFN_WV=8" "GE2E_ATTN_SUB=1" "GE2E_NO_LAST_CHAIN=1"; do
  env $sw timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16_train_step or fp32_train_step or golden" > gpurun_out/sw.log 2>&1 && echo "$sw ok: $(tail -1 gpurun_out/sw.log)" || { echo "$sw FAILED"; tail -15 gpurun_out/sw.log; }
done
