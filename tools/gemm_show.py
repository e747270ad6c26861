import sys, json
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_rf.txt"):
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['shape'], d['S'], d['bits'], d['fused_us'], d['dequant_matmul_us'], d['frac_of_2500_TF'])
    elif 'amdgpu.ids' not in l: print(l)
