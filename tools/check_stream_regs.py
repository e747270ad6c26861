"""Build-time check of csrc/ap_stream.hip: every plane-word request (inline-asm buffer_load_dwordx4 ... nt) of a kernel must write
one of RING * BITS * NH fixed register tuples -- a request site that got registers of its own means the compiler will copy them into
the slot's registers right behind the request, before the data has landed.  Usage:
    python tools/check_stream_regs.py [extra hipcc flags, e.g. -DST_MAXBITS=4]"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(flags):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "s.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                               "-fno-slp-vectorize", "--cuda-device-only", "-S", os.path.join(ROOT, "guidedquant_amd/csrc/ap_stream.hip"), "-o", out] + flags,
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    bad = 0
    for m in re.finditer(r"^(_ZN[^\n:]*ap_stream_kernelILi(\d)ELi(\d)ELi(\d)ELb(\d)E[^\n:]*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        bits = int(m.group(2))
        dst = collections.Counter(re.findall(r"buffer_load_dwordx4 (v\[\d+:\d+\]), v\d+, s\[\d+:\d+\], s\d+ offen nt", m.group(6)))
        want = 4 * bits  # RING slots x BITS planes (NH = 1)
        ok = len(dst) == want and max(dst.values()) <= 2
        bad += not ok
        print(f"bits={bits} pro={m.group(3)} npu={m.group(4)} psum={m.group(5)}: {len(dst)} destination tuples (want {want}), sites per tuple {sorted(set(dst.values()))} {'ok' if ok else 'MISMATCH'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
