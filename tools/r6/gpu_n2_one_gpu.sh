#!/bin/bash
# the N > 1 code path of bench.py (replicas line + pipeline_70b + tp_70b child groups) with both ranks on ONE GPU (gloo rendezvous);
# functional evidence only -- the numbers of such a run mean nothing
TAG=${1:-r06}
mkdir -p gpurun_out
export GQ_BENCH_LEG_TIMEOUT_S=${LEG_TIMEOUT:-240} PYTHONFAULTHANDLER=1 GQ_BENCH_ONE_GPU=1 GQ_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$SECONDS
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 \
  > gpurun_out/${TAG}_n2_one_gpu.json 2> gpurun_out/${TAG}_n2_one_gpu.err
echo "rc $? wall $((SECONDS - T0)) s"
tail -5 gpurun_out/${TAG}_n2_one_gpu.err
cat gpurun_out/${TAG}_n2_one_gpu.json | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'n_gpus', d['n_gpus']); print('pipeline_70b', d.get('pipeline_70b')); print('tp_70b', d.get('tp_70b')); print(d.get('distributed'))"
