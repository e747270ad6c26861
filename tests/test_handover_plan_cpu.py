"""CPU (host logic): gq_anyprec_handover_plan runs the library's own dispatch dry -- which launches of a decode step take the
statistics hand-over (include/gq_hip.h, round 5).  No device is touched (256 CUs are assumed without one)."""
from guidedquant_amd import _lib


def test_plan_follows_the_dispatch():
    L = _lib.lib()
    L.gq_set_ap_mode(0)
    try:
        # 8B 2-bit: the stream kernel's RMSNorm prologues read, the local-image kernel's residual epilogues write
        assert L.gq_anyprec_handover_plan(6144, 4096, 2, 1, 0) == 1
        assert L.gq_anyprec_handover_plan(28672, 4096, 2, 1, 4) == 1      # gate/up pair epilogue: nothing to write
        assert L.gq_anyprec_handover_plan(4096, 4096, 2, 0, 1) == 2
        assert L.gq_anyprec_handover_plan(4096, 14336, 2, 0, 1) == 2
        # 3 bits: wqkv runs the shared-image plane kernel (round 5: from 20 M weights behind the RMSNorm prologue) -- its prologue has
        # no reading form, its plain epilogue could write; wo / w2 write from two epilogue waves per block
        assert L.gq_anyprec_handover_plan(6144, 4096, 3, 1, 0) == 2
        assert L.gq_anyprec_handover_plan(4096, 4096, 3, 0, 1) == 2
        # 70B: wo on the stream kernel, w2 split along K over blocks -- neither has the in-epilogue form
        assert L.gq_anyprec_handover_plan(8192, 8192, 2, 0, 1) == 0
        assert L.gq_anyprec_handover_plan(8192, 28672, 2, 0, 1) == 0
        # bad shapes plan nothing
        assert L.gq_anyprec_handover_plan(0, 4096, 2, 1, 0) == 0 and L.gq_anyprec_handover_plan(4096, 4096, 9, 0, 1) == 0
        L.gq_set_ap_mode(1)
        assert L.gq_anyprec_handover_plan(6144, 4096, 2, 1, 0) == 0 and L.gq_anyprec_handover_plan(4096, 4096, 2, 0, 1) == 0
    finally:
        L.gq_set_ap_mode(-1)
