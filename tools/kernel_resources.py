"""Per-kernel register / LDS / scratch usage and VALU instruction counts from a device-only assembly listing.

    hipcc --offload-arch=gfx950 <flags of mosfhet_amd/build.py> --cuda-device-only -S -o /tmp/capi.s mosfhet_amd/csrc/capi.hip
    python tools/kernel_resources.py /tmp/capi.s [name filter]
"""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    text = open(path).read()
    meta = text[text.index(".amdgpu_metadata"):]
    kernels = []
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        blk = ".agpr_count:" + blk
        d = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
        kernels.append(d)
    dm = demangle([k["name"] for k in kernels])
    # instruction counts per function body
    bodies = {}
    for m in re.finditer(r"^(_Z\w+):\s*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    print("%-86s %5s %5s %5s %7s %7s %6s %6s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "valu", "f64", "ds"))
    for k in kernels:
        name = dm[k["name"]].replace("mosfhet::", "")
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        if filt and filt not in name:
            continue
        body = bodies.get(k["name"], "")
        ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
        valu = sum(1 for i in ins if i.startswith("v_"))
        f64 = sum(1 for i in ins if i.startswith("v_") and "f64" in i)
        ds = sum(1 for i in ins if i.startswith("ds_"))
        print("%-86s %5s %5s %5s %7s %7s %6d %6d %6d" % (name[:86], k.get("vgpr_count"), k.get("agpr_count"), k.get("sgpr_count"),
                                                     k.get("group_segment_fixed_size"), k.get("private_segment_fixed_size"), valu, f64, ds))


if __name__ == "__main__":
    main()
