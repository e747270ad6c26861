#!/bin/bash
# the N > 1 code path of bench.py with the ranks sharing the box's one GPU (numbers mean nothing for xGMI; what is compared is the
# host work per hop): 2 ranks, 8B model as the pipeline, default transport (gloo, staged through pinned host memory) vs GQ_PP_HOP=ipc
export GQ_BENCH_ONE_GPU=1 GQ_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for hop in p2p ipc; do
  echo "== GQ_PP_HOP=$hop"
  GQ_PP_HOP=$hop timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 100 --warmup 20 --parallel pp --model meta-llama/Meta-Llama-3.1-8B-Instruct --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | cut -c1-400
done
