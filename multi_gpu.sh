#!/bin/sh
# One rank per MI355X over RCCL (the counterpart of the reference's multi_gpu.sh:2, which uses torch.distributed.launch).
#   ./multi_gpu.sh [N_GPUS] [Hyper_Parameters.yaml]
# `python -m speaker_embedding_torch_amd.Train -hp <yaml>` with `Use_Multi_GPU: true` starts the same ranks by itself,
# and so does `python bench.py --gpus N`.
N=${1:-8}
HP=${2:-speaker_embedding_torch_amd/Hyper_Parameters.yaml}
HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 exec python -m torch.distributed.run --nnodes=1 --nproc-per-node="$N" \
    --master-addr 127.0.0.1 --master-port 54321 -m speaker_embedding_torch_amd.Train --hyper_parameters "$HP"
