"""guidedquant_amd -- MI355X (gfx950) implementation of GuidedQuant's quantized-linear decode path.

Only what the hot path needs: csrc/ (HIP kernels + the C ABI of include/gq_hip.h), and the host-side mirror of
the reference's operator surface (ap_gemv, plugin, APLinear, LUTGEMMLinear, AnyPrecisionLinear, ...).
"""
__version__ = "0.1.0"
