"""Static check of generated code for the compiler reordering that caused the "wrong units" of rounds 3 - 5 (experiments/README.md "Round 5"): LDS reads that the
source places IN FRONT of a workgroup barrier, emitted BEHIND it.

Every wave-level exchange of the transforms is  ds_write ... ; wave barrier ; ds_read ...  (negacyclic_fft.h: wave_lds_sync between the writes and the reads of ONE
wavefront), and the reads always stand in front of the next workgroup barrier in the source.  In a device listing (hipcc -S) the wave barrier survives as the
comment line "; wave barrier"; so: behind every such line that follows LDS writes, the next LDS access (in listing order, across branches and labels) must come BEFORE
the next s_barrier.  A listing that violates this has had its reads moved across the barrier (LLVM's machine sinking does not treat S_BARRIER as a store).
The two-wavefront exchanges (ds_write ; s_barrier ; ds_read ; s_barrier: both wavefronts write, meet, read each other's half, meet again) are held the same way:
inside one basic block, LDS writes followed by TWO barriers with no LDS read in between mean the reads went behind the barrier that releases the buffer.

    python tools/check_lds_barriers.py listing.s [...]        exit status 1 and one line per violation
    build_and_check()                                          compiles tools/ab/ep_ab.hip in the forms that used to fail and checks them (CPU, seconds)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(text):
    """[(kernel, line number, what)] for every wave-level exchange whose reads were emitted behind a workgroup barrier / a branch"""
    out, kernel, pending, wrote = [], None, None, False
    cross, cross_wrote = None, False      # the two-wavefront A <-> B exchanges: ds_write ; s_barrier ; ds_read ; s_barrier
    for no, raw in enumerate(text.split("\n"), 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+|k_raw|k_fixed):", line)
        if m:
            kernel, pending, wrote, cross, cross_wrote = m.group(1), None, False, None, False
            continue
        if line.startswith("; wave barrier"):
            pending = no if wrote else None       # an exchange is open: writes done, reads due
            cross_wrote = False                   # (a wave-level exchange, not a cross-wavefront one)
            continue
        if re.match(r"^\.?L?BB\d+_\d+:", line):  # a new basic block: another path may arrive here (a team that sits a transform out walks barriers without having written)
            cross, cross_wrote = None, False
            continue
        op = line.split()[0] if line and not line.startswith((";", ".")) else ""
        if op.startswith("ds_write"):
            wrote = cross_wrote = True
            cross = None
            if pending:
                pending = None                       # (a new exchange's writes: the previous one had no reads of its own, e.g. a parked transform)
        elif op.startswith("ds_read"):
            wrote, pending, cross, cross_wrote = False, None, None, False
        elif op == "s_barrier":
            if pending:                              # (reads behind a branch or a label alone are harmless: one wavefront, DS instructions execute in order)
                out.append((kernel, no, "the reads of the exchange opened at line %d come behind this s_barrier" % pending))
                pending = None
            if cross:                                # writes ; barrier ; NO read ; barrier in one basic block: the reads of a cross-wavefront exchange were moved behind the
                out.append((kernel, no, "no LDS read between the barrier at line %d (behind LDS writes) and this one: a cross-wavefront exchange's reads were moved" % cross))   # barrier that frees its buffer
                cross = None
            elif cross_wrote:
                cross, cross_wrote = no, False
        elif op == "s_endpgm":
            pending, wrote, cross, cross_wrote = None, False, None, False
    return out


FORMS = [("N = 2048, l = 1, pipelined loop (the build that failed)", ["-DAB_N=2048", "-DAB_L=1", "-DAB_BG=23", "-DAB_FORM=2"]),
         ("N = 4096, l = 1, pipelined loop", ["-DAB_N=4096", "-DAB_L=1", "-DAB_BG=22", "-DAB_FORM=2"]),
         ("N = 2048, l = 4, rows in pairs, pipelined loop (lvl2 production)", ["-DAB_N=2048", "-DAB_L=4", "-DAB_BG=9", "-DAB_LTW"]),
         ("N = 2048, l = 4, plain loop", ["-DAB_N=2048", "-DAB_L=4", "-DAB_BG=9", "-DAB_LTW", "-DAB_FORM=1"])]


def build_and_check(forms=FORMS):
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, flags in forms:
            s = os.path.join(tmp, "k.s")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-Wno-comment", "-S", "--cuda-device-only"] + flags +
                                  [os.path.join(ROOT, "tools", "ab", "ep_ab.hip"), "-o", s], stderr=subprocess.DEVNULL)
            text = open(s).read()
            assert "; wave barrier" in text and "s_barrier" in text, "the listing of `%s` has no barriers to check" % name
            bad += [(name,) + v for v in check(text)]
    return bad


if __name__ == "__main__":
    found = []
    for path in sys.argv[1:]:
        found += [(path,) + v for v in check(open(path).read())]
    if not sys.argv[1:]:
        found = build_and_check()
    for v in found:
        print("%s: %s line %d: %s" % v)
    sys.exit(1 if found else 0)
