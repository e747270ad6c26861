"""CPU: no kernel of the built library spills vector registers or uses scratch memory (tools/kernel_resources.py reads the metadata
notes of the gfx950 code objects in guidedquant_amd/csrc/*.o).  Round 4 shipped exact-mode GEMV instances (ring depth D >= 2 at 2 / 3
bits, D = 4 at 4 bits) with up to 193 spilled VGPRs behind the GQ_AP_D knob; the dispatcher can no longer pick a spilling instance
(csrc/ap_gemv.hip::pick_quad_cfg) and none is compiled.  SGPR spills go to VGPR lanes (v_writelane), not to memory, and are allowed."""
import glob
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_kernel_spills_vgprs_or_uses_scratch():
    from guidedquant_amd import _lib
    import kernel_resources as kr
    _lib.build()  # incremental: a no-op when the objects are current
    objs = sorted(glob.glob(os.path.join(ROOT, "guidedquant_amd", "csrc", "*.o")))
    assert len(objs) >= 12
    seen, bad = 0, []
    for obj in objs:
        try:
            ks = kr.kernels_of(obj)
        except Exception:  # host-only translation unit (cpu_twins.o): no .hip_fatbin section
            assert os.path.basename(obj) == "cpu_twins.o", obj
            continue
        seen += len(ks)
        bad += [(os.path.basename(obj), k[0], k[4], k[6]) for k in ks if k[4] or k[6]]
    assert seen >= 250, seen  # (every template instance of the library)
    assert not bad, bad


def test_exact_gemv_instances_fit_their_occupancy():
    """ap_gemv_quad_kernel<BITS, D, PRO, INREG>: D = 1 at 2 / 3 bits is built for 3 waves per SIMD (<= 168 VGPRs), the in-register
    reduction instances (INREG, D = 1) for 4 waves at 2 bits (<= 128) and 3 at 3 / 4 bits, everything else for 2."""
    import kernel_resources as kr
    ks = [k for k in kr.kernels_of(os.path.join(ROOT, "guidedquant_amd", "csrc", "ap_gemv.o")) if "ap_gemv_quad_kernel" in k[0]]
    assert len(ks) == 42  # bits 2, 3: D 1..4; bits 4: D 1..3; x 3 prologues; + INREG: 3 bit widths x 3 prologues
    for k in ks:
        bits, d, inreg = int(k[0].split("ILi")[1][0]), int(k[0].split("ELi")[1][0]), "ELb1E" in k[0]
        assert k[1] <= ((128 if bits == 2 else 168) if inreg else (168 if bits <= 3 and d == 1 else 256)), k
