import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from guidedquant_amd import _lib
os.environ['GQ_PL_MIN_MWEIGHTS']='0'
L=_lib.lib(); L.gq_set_ap_mode(0)
d=torch.device("cuda:0")
for (N,K) in [(4096,4096),(8192,8192)]:
    torch.manual_seed(N+K)
    q=torch.randint(-2**31,2**31-1,(2,N,K//32),dtype=torch.int32,device=d); lut=(torch.randn(N,4,device=d)*0.02).half()
    x=torch.randn(K,device=d).half(); res=torch.randn(N,device=d).half()
    outs={}
    for st in (0,3):
        os.environ["GQ_ST"]=str(st); L.gq_reset_env_cache()
        o1=torch.zeros(N,dtype=torch.float16,device=d); o2=torch.zeros(N,dtype=torch.float16,device=d)
        assert L.gq_anyprec_gemv_fused(x.data_ptr(),o1.data_ptr(),q.data_ptr(),lut.data_ptr(),N,K,2,None,0.0,None,0,None)==0
        assert L.gq_anyprec_gemv_fused(x.data_ptr(),o2.data_ptr(),q.data_ptr(),lut.data_ptr(),N,K,2,None,0.0,res.data_ptr(),1,None)==0
        torch.cuda.synchronize()
        exp=(res+o1)
        bad=(exp.view(torch.int16)!=o2.view(torch.int16)).nonzero().view(-1)
        print(N,K,"st",st,"resid mismatches",bad.numel(), bad[:8].tolist(), (o2.float()-exp.float()).abs().max().item())
        outs[st]=o1
    print("  st3 vs st0 maxdiff", (outs[0].float()-outs[3].float()).abs().max().item())
