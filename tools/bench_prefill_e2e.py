"""Prefill of the whole Llama-3.1-8B-shaped model (random init, Any-Precision 2/3/4-bit): `model(x, input_pos)` over an S-token
prompt -- the call generate() makes before the decode loop (inference/generate.py:82-93 `prefill`) -- with the seq_len > 1
linears through the fused prefill GEMM (default dispatch) and through the reference's two steps (GQ_PREFILL_FUSED=0:
anyprec_dequant + torch.matmul), and the HIP prompt pass (`Transformer.prefill_native`: what generate() takes; logits of the last
token only).  ms per prompt, HIP events (the host-side launch overhead is inside: the stream is idle when each call starts),
best of 3."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidedquant_amd.generate import load_model  # noqa: E402

if __name__ == "__main__":
    d = torch.device("cuda:0")
    bits_list = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2]
    for bits in bits_list:
        torch.manual_seed(0)
        m = load_model("meta-llama/Meta-Llama-3.1-8B", d, "ap", bits, random_init=True)
        m.setup_caches(1, 2048 + 8)
        for S in (128, 512, 2048):
            x = torch.randint(0, 128000, (1, S), dtype=torch.int32, device=d)
            pos = torch.arange(S, dtype=torch.int32, device=d)
            res = {}
            for mode in ("native", "module", "module_two_steps"):
                os.environ["GQ_PREFILL_FUSED"] = "0" if mode == "module_two_steps" else "auto"
                fn = (lambda: m.prefill_native(x, pos, start=0, last_only=True)) if mode == "native" else (lambda: m(x, pos))
                with torch.no_grad():
                    fn()
                    torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        lg = fn()
                        e1.record()
                        e1.synchronize()
                        best = min(best, e0.elapsed_time(e1))
                res[mode] = best
                assert torch.isfinite(lg.float()).all()
            os.environ.pop("GQ_PREFILL_FUSED", None)
            print(json.dumps({"model": "Llama-3.1-8B", "bits": bits, "prompt_tokens": S, "prefill_ms_native_prompt_pass": round(res["native"], 3),
                              "prefill_ms_module_forward_fused_gemm": round(res["module"], 3),
                              "prefill_ms_module_forward_reference_two_steps": round(res["module_two_steps"], 3),
                              "speedup_vs_reference_steps": round(res["module_two_steps"] / res["native"], 2),
                              "prompt_tokens_per_s": round(S / res["native"] * 1e3, 0)}), flush=True)
        del m
        torch.cuda.empty_cache()
