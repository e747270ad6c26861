#!/bin/bash
# the N > 1 code paths of bench.py with the ranks sharing the box's ONE GPU (functional check only; the numbers mean nothing)
out=gpurun_out/r5_call37.txt; mkdir -p gpurun_out; : > $out
export GQ_BENCH_ONE_GPU=1 GQ_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "### replicas line + pipeline_70b sub-record, 2 ranks (driver form)"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 2>gpurun_out/r5_n2_err.txt | tail -1 > gpurun_out/r5_n2_line.json; python -c "import json; d=json.load(open(\"gpurun_out/r5_n2_line.json\")); print(d[\"value\"], d[\"n_gpus\"], json.dumps(d.get(\"pipeline_70b\"))[:600], json.dumps(d.get(\"distributed\")))"
tail -3 gpurun_out/r5_n2_err.txt | cut -c1-300
echo "### --parallel pp, 8B as the pipeline, p2p / ipc hops"
for hop in p2p ipc; do
  echo "== GQ_PP_HOP=$hop"
  GQ_PP_HOP=$hop timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 100 --warmup 20 --parallel pp --model meta-llama/Meta-Llama-3.1-8B-Instruct --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | cut -c1-500
done
} >> $out 2>&1
