"""Per-kernel register / scratch figures of the BUILT code objects (guidedquant_amd/csrc/*.o): the gfx950 code object is taken out of
each fat object (llvm-objcopy + clang-offload-bundler) and its metadata notes are read (llvm-readelf).  No GPU needed.
    python tools/kernel_resources.py            # every kernel that spills or uses scratch
    python tools/kernel_resources.py --all      # every kernel"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    """[(mangled name, vgprs, agprs, sgprs, spilled vgprs, spilled sgprs, scratch bytes, lds bytes)] of one fat object"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([f"{BIN}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.devnull])
        subprocess.check_call([f"{BIN}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.run([f"{BIN}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        def f(key, blk=blk):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else "0"
        agpr = re.match(r"\s*(\d+)", blk).group(1)
        out.append((f("name"), int(f("vgpr_count")), int(agpr), int(f("sgpr_count")), int(f("vgpr_spill_count")), int(f("sgpr_spill_count")),
                    int(f("private_segment_fixed_size")), int(f("group_segment_fixed_size"))))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [re.sub(r"\(anonymous namespace\)::", "", n) for n in r.stdout.split("\n")]


def main(argv):
    show_all = "--all" in argv
    objs = sorted(glob.glob(os.path.join(ROOT, "guidedquant_amd", "csrc", "*.o")))
    bad = 0
    for obj in objs:
        if b".hip_fatbin" not in subprocess.run([f"{BIN}/llvm-readelf", "-S", obj], capture_output=True).stdout:
            continue  # host-only translation unit
        ks = kernels_of(obj)
        names = demangle([k[0] for k in ks])
        for k, n in zip(ks, names):
            spills = k[4] or k[5] or k[6]
            bad += bool(spills)
            if show_all or spills:
                print(f"{os.path.basename(obj):14s} v={k[1]:3d} a={k[2]:3d} s={k[3]:3d} vspill={k[4]:3d} sspill={k[5]:2d} scratch={k[6]:4d} lds={k[7]:6d}  {n[:110]}")
    print(f"{bad} kernels with spills / scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
