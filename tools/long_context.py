"""decode tokens/s of the headline model at a long context (bench.long_context_record): python tools/long_context.py [start] [steps];
GQ_ATTN_GQA=0 / GQ_ATTN_SPLIT=n select the attention form"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
start = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
print(json.dumps(bench.long_context_record(torch.device("cuda:0"), start, steps)))
