"""Kernel table of the built library: every kernel instantiation in mosfhet_amd/libmosfhet_hip.so with its registers, LDS and scratch, read from the code
object's own metadata (the .hip_fatbin section -> gfx950 code object -> AMDGPU notes).  Needs only the LLVM tools of the ROCm image; no GPU.

    python tools/kernel_table.py [--scratch] [name filter]        (--scratch: only kernels with a private segment)

tests/test_host_and_abi.py uses table() to hold the build to what the launcher assumes about it (capi.hip: ep_go).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def table(lib=None):
    """[{name (demangled), vgpr, agpr, sgpr, lds, scratch, max_threads}] of every kernel of the library"""
    lib = lib or os.path.join(ROOT, "mosfhet_amd", "libmosfhet_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "lib.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        d = dict(re.findall(r"\.(\w+):\s+(\S+)", ".agpr_count:" + blk))
        kernels.append(d)
    names = [k["name"] for k in kernels]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    out = []
    for k, n in zip(kernels, dem):
        n = re.sub(r"\(.*", "", n.replace("mosfhet::", "").replace("void ", ""))
        out.append(dict(name=n, vgpr=int(k["vgpr_count"]), agpr=int(k["agpr_count"]), sgpr=int(k["sgpr_count"]), lds=int(k["group_segment_fixed_size"]),
                        scratch=int(k["private_segment_fixed_size"]), max_threads=int(k["max_flat_workgroup_size"])))
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--scratch"]
    rows = [r for r in table() if (not args or args[0] in r["name"]) and ("--scratch" not in sys.argv or r["scratch"])]
    print("%-100s %5s %5s %7s %7s" % ("kernel", "vgpr", "sgpr", "lds", "scratch"))
    for r in sorted(rows, key=lambda r: r["name"]):
        print("%-100s %5d %5d %7d %7d" % (r["name"][:100], r["vgpr"], r["sgpr"], r["lds"], r["scratch"]))
    print("%d kernels, %d with scratch" % (len(rows), sum(1 for r in rows if r["scratch"])))
