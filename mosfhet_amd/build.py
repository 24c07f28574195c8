"""Builds the in-tree native library mosfhet_amd/libmosfhet_hip.so for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container too.  The .so is
git-ignored but travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmosfhet_hip.so")

HIP_SOURCES = ["capi.hip"]
HOST_C_SOURCES = ["host/mosfhet_compat.c", "host/mosfhet_compat_dft.c", "host/mosfhet_compat_multi.c", "host/mosfhet_compat_legacy.c", "host/mosfhet_compat_extra.c", "host/mosfhet_compat_vec.c", "host/csprng.c"]
DEPS = ["negacyclic_fft.h", "bootstrap_kernels.h", "general_kernels.h", "keyswitch_kernels.h", "keyswitch_words_kernels.h", "ks_words_asm.inc", "ext_kernels.h", "unfold_kernels.h", "keygen_kernels.h", "capi_ext.inc", "capi_dft.inc", "capi_vec.inc", "../../include/mosfhet_hip.h",
        "../../include/mosfhet_compat.h", "../../include/mosfhet.h", "host/compat_internal.h", "../build.py", "../../tools/check_lds_barriers.py"]

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
               "-Wno-comment"]
CC_FLAGS = ["-O3", "-mavx2", "-std=gnu11", "-fPIC", "-ffp-contract=off", "-Wall"]  # -O3 -mavx2: vectorised host key generation


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


STAMP = LIB + ".srchash"   # content hash of the sources the library was built from (travels with the .so)


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in sorted(HIP_SOURCES + HOST_C_SOURCES + DEPS):
        path = os.path.join(CSRC, f)
        if os.path.exists(path):
            h.update(f.encode())
            with open(path, "rb") as fh:
                h.update(fh.read())
    h.update(os.environ.get("MOSFHET_HIPCC_EXTRA", "").encode())
    return h.hexdigest()


def _stale():
    """Stale = the sources' CONTENT differs from what the library was built from (mtimes do not survive the copy to a GPU box,
    and a rebuild there by every rank at once would race)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != _source_hash()


def build(force=False, verbose=False):
    """Compile csrc/*.hip (device + C ABI) and csrc/host/*.c (MOSFHET-compatible host layer) into one .so."""
    if not force and not _stale():
        return LIB
    import fcntl
    lock = open(LIB + ".lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)   # one builder at a time (torch.distributed.run starts one process per GPU)
    try:
        if not force and not _stale():
            return LIB
        return _build_locked(verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _check_listing(objdir, src):
    """the device listing -save-temps left behind: no LDS read behind the barrier its exchange stands in front of; then the temporaries go (tens of MB)"""
    import glob
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    import check_lds_barriers
    stem = os.path.join(objdir, os.path.splitext(src)[0])
    found = glob.glob(stem + "-hip-*gfx950*.s")
    if len(found) != 1:
        raise RuntimeError("the gfx950 device listing of %s (-save-temps=obj) was not found next to the object (%s-hip-*gfx950*.s matched %d files): the LDS-barrier check "
                           "of the build cannot run -- MOSFHET_HIPCC_EXTRA must not add or change --offload-arch" % (src, stem, len(found)))
    with open(found[0]) as fh:
        text = fh.read()
    bad = check_lds_barriers.check(text)
    with open(os.path.join(objdir, "lds_barrier_check.txt"), "w") as fh:
        fh.write("%s: %d kernels, %d wave-level exchanges, %d workgroup barriers, %d violations\n" % (src, text.count(".amdhsa_kernel "), text.count("; wave barrier"),
                                                                                                   text.count("\ts_barrier"), len(bad)))
        for v in bad:
            fh.write("%s line %d: %s\n" % v)
    for tmp in glob.glob(stem + "-hip-*") + glob.glob(stem + "-host-*") + glob.glob(stem + ".hip-hip-*"):
        os.remove(tmp)
    if bad:
        raise RuntimeError("the compiler emitted LDS reads behind a workgroup barrier (%d places, first: %s line %d): see %s" % (
            len(bad), bad[0][0], bad[0][1], os.path.join(objdir, "lds_barrier_check.txt")))


def _build_locked(verbose):
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in HOST_C_SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = ["gcc"] + CC_FLAGS + ["-I", os.path.join(HERE, "..", "include"), "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    for src in HIP_SOURCES:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        # -save-temps=obj leaves the DEVICE LISTING of the very code that is linked next to the object: it is checked for LDS reads emitted behind a workgroup
        # barrier (tools/check_lds_barriers.py: the compiler reordering behind the wrong units of rounds 3 - 5) before the library is allowed to exist
        cmd = [_hipcc()] + HIPCC_FLAGS + os.environ.get("MOSFHET_HIPCC_EXTRA", "").split() + ["-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        _check_listing(objdir, os.path.basename(src))
        objs.append(obj)
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    with open(STAMP, "w") as fh:
        fh.write(_source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
