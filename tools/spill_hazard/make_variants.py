"""Assembly-level bisection of the round-3 wrong-unit failure of the software-pipelined external-product loop on two-wavefront rings.

The failing build (tools/ab/ep_ab.hip, N = 2048, l = 1, Bg_bit = 23, -DAB_FORM=2: 28 bytes of scratch) is compiled to device assembly ONCE; every
variant below is a textual patch of that one listing (same schedule, same registers, same waits elsewhere), assembled with clang/lld and wrapped into the
host side of ep_ab.hip, so `tools/ab/run_ab.py run_ep` loads it like any other variant (tools/ab/_build/ep_<name>.so).

    python tools/spill_hazard/make_variants.py            (CPU container; hipcc cross-compiles)
    python tools/ab/run_ab.py run_ep 16384 3 1024 set2 <names>     (GPU box)

The anchors are the compiler's own lines of THIS build (ROCm 7.2.0); the script stops if one is not found.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "tools", "ab", "_build")
WORK = os.path.join(OUT, "spill_hazard")
LLVM = "/opt/rocm/lib/llvm/bin"
SRC = os.path.join(ROOT, "tools", "ab", "ep_ab.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value", "-Wno-comment", "-DAB_N=2048", "-DAB_L=1", "-DAB_BG=23", "-DAB_FORM=2"]

SPILL_WAIT = "\ts_waitcnt vmcnt(13)\n\tscratch_store_dwordx2 off, v[46:47], off offset:8 ; 8-byte Folded Spill\n"
SPILL_LAST = "\tscratch_store_dwordx2 off, v[0:1], off  ; 8-byte Folded Spill\n"
RELOAD_0 = "\tscratch_load_dwordx2 v[50:51], off, off ; 8-byte Folded Reload\n"
RELOAD_1 = "\tscratch_load_dwordx2 v[52:53], off, off offset:8 ; 8-byte Folded Reload\n"
LDS_BYTES = 18432          # the kernel's own LDS; the mirror area sits behind it (16 bytes per thread, v192 = 16 * threadIdx.x)
PROLOGUE = "; %bb.0:\n\ts_load_dword s44, s[0:1], 0x24\n"


def sh(cmd):
    subprocess.check_call(cmd)


def once(text, old, new):
    assert text.count(old) == 1, "anchor not found exactly once: %r (%d)" % (old, text.count(old))
    return text.replace(old, new)


def lds_size(text, size):
    text = once(text, "\t\t.amdhsa_group_segment_fixed_size %d\n" % LDS_BYTES, "\t\t.amdhsa_group_segment_fixed_size %d\n" % size)
    return re.sub(r"(\.group_segment_fixed_size: )%d(\n(?:.*\n){1,12}?\s+\.name:\s+_ZN7mosfhet23external_product_kernel)" % LDS_BYTES, r"\g<1>%d\2" % size, text, count=1)


def variants(base, only=None):
    v = {"asm_base": base}
    v["wait_after_spill"] = once(base, SPILL_LAST, SPILL_LAST + "\ts_waitcnt vmcnt(0)\n")
    v["wait_all_before_spill"] = once(base, SPILL_WAIT, SPILL_WAIT.replace("vmcnt(13)", "vmcnt(0)"))
    v["wait_after_reload"] = once(base, RELOAD_1, RELOAD_1 + "\ts_waitcnt vmcnt(0)\n")
    v["wait_before_reload"] = once(base, RELOAD_0, "\ts_waitcnt vmcnt(0)\n" + RELOAD_0)
    v["inv_before_reload"] = once(base, RELOAD_0, "\ts_waitcnt vmcnt(0)\n\tbuffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)\n" + RELOAD_0)
    t = once(base, RELOAD_0, RELOAD_0.replace("off, off ;", "off, off sc0 sc1 ;"))
    v["reload_sc0sc1"] = once(t, RELOAD_1, RELOAD_1.replace("offset:8 ;", "offset:8 sc0 sc1 ;"))
    t = once(base, SPILL_LAST, SPILL_LAST.replace("off  ;", "off sc0 sc1 ;"))
    v["spill_sc0sc1"] = once(t, SPILL_WAIT, SPILL_WAIT.replace("offset:8 ;", "offset:8 sc0 sc1 ;"))
    # the two spilled words in LDS instead of scratch (same places, same registers)
    t = once(base, SPILL_WAIT, "\ts_waitcnt vmcnt(13)\n\tds_write_b64 v192, v[46:47] offset:%d\n" % (LDS_BYTES + 8))
    t = once(t, SPILL_LAST, "\tds_write_b64 v192, v[0:1] offset:%d\n\ts_waitcnt lgkmcnt(0)\n" % LDS_BYTES)
    t = once(t, RELOAD_0, "\tds_read_b64 v[50:51], v192 offset:%d\n" % LDS_BYTES)
    t = once(t, RELOAD_1, "\tds_read_b64 v[52:53], v192 offset:%d\n\ts_waitcnt lgkmcnt(0)\n" % (LDS_BYTES + 8))
    v["lds_instead"] = lds_size(t, LDS_BYTES + 2048)
    # scratch kept, mirrored in LDS and compared at the reload; mismatches are counted and sampled into the buffer passed as `in0` (unused by this instantiation):
    #   word 0: lanes that saw a mismatch; from byte 64: per thread 32 bytes {scratch word 0, LDS word 0, scratch word 1, LDS word 1}
    for fix in (False, True):
        t = once(base, PROLOGUE, PROLOGUE + "\ts_load_dwordx2 s[60:61], s[0:1], 0x38\n\ts_lshl_b32 s62, s2, 12\n")
        t = once(t, SPILL_LAST, SPILL_LAST + "\tds_write_b64 v192, v[46:47] offset:%d\n\tds_write_b64 v192, v[0:1] offset:%d\n\ts_waitcnt lgkmcnt(0)\n" % (LDS_BYTES + 8, LDS_BYTES))
        check = ("\ts_waitcnt vmcnt(0)\n"
                 "\tds_read_b64 v[42:43], v192 offset:%d\n\tds_read_b64 v[44:45], v192 offset:%d\n\ts_waitcnt lgkmcnt(0)\n" % (LDS_BYTES, LDS_BYTES + 8) +
                 "\tv_cmp_ne_u64_e32 vcc, v[42:43], v[50:51]\n\tv_cmp_ne_u64_e64 s[64:65], v[44:45], v[52:53]\n\ts_or_b64 vcc, vcc, s[64:65]\n"
                 "\ts_and_saveexec_b64 s[66:67], vcc\n\ts_cbranch_execz .LSH_skip\n"
                 "\tv_mov_b32_e32 v46, 0\n\tv_mov_b32_e32 v47, 1\n\tglobal_atomic_add v46, v47, s[60:61]\n"
                 "\tv_lshl_add_u32 v48, v192, 1, s62\n"
                 "\tglobal_store_dwordx2 v48, v[50:51], s[60:61] offset:64\n\tglobal_store_dwordx2 v48, v[42:43], s[60:61] offset:72\n"
                 "\tglobal_store_dwordx2 v48, v[52:53], s[60:61] offset:80\n\tglobal_store_dwordx2 v48, v[44:45], s[60:61] offset:88\n"
                 "\ts_waitcnt vmcnt(0)\n" +
                 ("\tv_mov_b64_e32 v[50:51], v[42:43]\n\tv_mov_b64_e32 v[52:53], v[44:45]\n" if fix else "") +
                 ".LSH_skip:\n\ts_or_b64 exec, exec, s[66:67]\n")
        t = once(t, RELOAD_1, RELOAD_1 + check)
        v["mirror_fix" if fix else "mirror"] = lds_size(t, LDS_BYTES + 2048)
    # ---- second round: where does the timing dependence sit? ----
    LOOP = ".LBB1_3:                                ; =>This Inner Loop Header: Depth=1\n"
    v["barrier_loop_top"] = once(base, LOOP, LOOP + "\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n")
    v["drain_barrier_loop_top"] = once(base, LOOP, LOOP + "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n")
    v["lds_40k"] = lds_size(base, 40960)
    v["lds_80k"] = lds_size(base, 81920)
    a, b = base.index("\n_ZN7mosfhet23external_product_kernel"), base.index("\t.section\t.rodata", base.index("\n_ZN7mosfhet23external_product_kernel"))
    body = base[a:b]

    def every(pattern, extra):
        return base[:a] + re.sub(pattern, lambda m: m.group(0) + extra, body, flags=re.M) + base[b:]
    v["vmcnt0_at_barriers"] = base[:a] + body.replace("\ts_barrier\n", "\ts_waitcnt vmcnt(0)\n\ts_barrier\n") + base[b:]
    v["zero_all"] = every(r"^\t(?:global_load|global_store|scratch_load|scratch_store|ds_read|ds_write)\S* .*\n", "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n")
    v["zero_lgkm"] = every(r"^\t(?:ds_read|ds_write)\S* .*\n", "\ts_waitcnt lgkmcnt(0)\n")
    v["zero_vm"] = every(r"^\t(?:global_load|global_store|scratch_load|scratch_store)\S* .*\n", "\ts_waitcnt vmcnt(0)\n")
    v["zero_vm_loads"] = every(r"^\t(?:global_load|scratch_load)\S* .*\n", "\ts_waitcnt vmcnt(0)\n")
    v["zero_vm_ct_loads"] = every(r"^\tglobal_load_dwordx2 .* nt\n", "\ts_waitcnt vmcnt(0)\n")
    v["zero_vm_key_loads"] = every(r"^\tglobal_load_dwordx4 .*\n", "\ts_waitcnt vmcnt(0)\n")
    # ---- third round: the key-row loads (global_load_dwordx4) are what has to be drained; which, and is it the issue or the wait? ----
    lines = base.split("\n")
    hdr, mid, end = lines.index(LOOP.rstrip("\n")), next(i for i, x in enumerate(lines) if x.startswith(".LBB1_5:")), next(i for i, x in enumerate(lines) if x.startswith(".LBB1_9:"))

    def in_range(lo, hi, pattern, extra, repl=None):
        out = list(lines)
        for i in range(lo, hi):
            if re.match(pattern, out[i]):
                out[i] = (repl(out[i]) if repl else out[i]) + (("\n" + extra) if extra else "")
        return "\n".join(out)
    v["drain_keys_rows_a"] = in_range(hdr, mid, r"\tglobal_load_dwordx4 ", "\ts_waitcnt vmcnt(0)")
    v["drain_keys_rows_b"] = in_range(mid, end, r"\tglobal_load_dwordx4 ", "\ts_waitcnt vmcnt(0)")
    v["full_waits_only"] = in_range(hdr - 20, end, r"\ts_waitcnt vmcnt\(\d+\)", None, lambda x: re.sub(r"vmcnt\(\d+\)", "vmcnt(0)", x))
    v["full_waits_rows_a"] = in_range(hdr - 20, mid, r"\ts_waitcnt vmcnt\(\d+\)", None, lambda x: re.sub(r"vmcnt\(\d+\)", "vmcnt(0)", x))
    v["full_waits_rows_b"] = in_range(mid, end, r"\ts_waitcnt vmcnt\(\d+\)", None, lambda x: re.sub(r"vmcnt\(\d+\)", "vmcnt(0)", x))
    v["no_nt"] = in_range(0, end, r"\tglobal_(?:load|store)_dwordx2 .* nt$", None, lambda x: x[:-3])
    v["keys_nt"] = in_range(0, end, r"\tglobal_load_dwordx4 ", None, lambda x: x.rstrip() + " nt")
    v["keys_sc1"] = in_range(0, end, r"\tglobal_load_dwordx4 ", None, lambda x: x.rstrip() + " sc1")
    # ---- fourth round: full waits do not help, draining each key-row load does: is it time between a key load's issue and what follows it? ----
    v["nop4_after_key_loads"] = in_range(mid, end, r"\tglobal_load_dwordx4 ", "\ts_nop 3")
    v["nop16_after_key_loads"] = in_range(mid, end, r"\tglobal_load_dwordx4 ", "\ts_nop 15")
    v["nop64_after_key_loads"] = in_range(mid, end, r"\tglobal_load_dwordx4 ", "\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15")
    v["sleep_after_key_loads"] = in_range(mid, end, r"\tglobal_load_dwordx4 ", "\ts_sleep 8")
    v["lgkm0_after_key_loads"] = in_range(mid, end, r"\tglobal_load_dwordx4 ", "\ts_waitcnt lgkmcnt(0)")
    # ---- fifth round: are the key-row words WRONG when they are used?  Six key quads of the first row group are copied to AGPRs right behind the compiler's
    # own wait for them; at the loop latch (everything drained) the same addresses are read again and compared.  Mismatching lanes are counted in word 0 of the
    # debug buffer (passed as in0) and sampled: per thread 64 bytes from byte 64 on: {used quad, true quad, quad index + 1, unit counter}.
    # (24 AGPRs: 280 registers per wavefront, one wavefront per SIMD instead of two.)
    snap = [("\ts_waitcnt vmcnt(9)\n\tv_fma_f64 v[56:57], v[48:49], v[44:45], 0\n", [(0, 42)]),
            ("\ts_waitcnt vmcnt(8)\n\tv_fma_f64 v[118:119], v[60:61], v[50:51], 0\n", [(4, 50)]),
            ("\ts_waitcnt vmcnt(6)\n\tv_fma_f64 v[52:53], v[60:61], v[92:93], 0\n", [(8, 82), (12, 90)]),
            ("\ts_waitcnt vmcnt(4)\n\tv_fma_f64 v[42:43], v[114:115], v[122:123], 0\n", [(16, 94), (20, 122)])]
    t = once(base, PROLOGUE, PROLOGUE + "\ts_load_dwordx2 s[60:61], s[0:1], 0x38\n\ts_lshl_b32 s62, s2, 13\n")
    for anchor, quads in snap:
        first, rest = anchor.split("\n", 1)
        copies = "".join("\tv_accvgpr_write_b32 a%d, v%d\n" % (a + i, r + i) for a, r in quads for i in range(4))
        t = once(t, anchor, first + "\n" + copies + rest)
    offs = [0, 2048, 0x4000, 0x4000 + 2048, 0x1000, 0x5000]
    chk = "\ts_waitcnt vmcnt(0)\n\tv_lshl_add_u64 v[42:43], s[12:13], 0, v[192:193]\n\ts_mov_b64 s[64:65], 0\n\tv_lshl_add_u32 v1, v192, 2, s62\n"
    for q, off in enumerate(offs):
        chk += ("\tv_add_co_u32_e32 v48, vcc, 0x%x, v42\n\ts_nop 1\n\tv_addc_co_u32_e32 v49, vcc, 0, v43, vcc\n\tglobal_load_dwordx4 v[44:47], v[48:49], off\n"
                "\ts_waitcnt vmcnt(0)\n\ts_mov_b64 s[66:67], 0\n" % off)
        for i in range(4):
            chk += "\tv_accvgpr_read_b32 v0, a%d\n\ts_nop 0\n\tv_cmp_ne_u32_e32 vcc, v0, v%d\n\ts_or_b64 s[66:67], s[66:67], vcc\n" % (4 * q + i, 44 + i)
        chk += "\ts_or_b64 s[64:65], s[64:65], s[66:67]\n\ts_and_saveexec_b64 s[68:69], s[66:67]\n\ts_cbranch_execz .LVK_%d\n" % q
        for i in range(4):
            chk += "\tv_accvgpr_read_b32 v0, a%d\n\ts_nop 0\n\tglobal_store_dword v1, v0, s[60:61] offset:%d\n" % (4 * q + i, 64 + 4 * i)
        chk += ("\tglobal_store_dwordx4 v1, v[44:47], s[60:61] offset:80\n\tv_mov_b32_e32 v0, %d\n\tglobal_store_dword v1, v0, s[60:61] offset:96\n"
                "\tv_mov_b32_e32 v0, s2\n\tglobal_store_dword v1, v0, s[60:61] offset:100\n\ts_waitcnt vmcnt(0)\n.LVK_%d:\n\ts_or_b64 exec, exec, s[68:69]\n" % (q + 1, q))
    chk += ("\ts_and_saveexec_b64 s[68:69], s[64:65]\n\ts_cbranch_execz .LVK_end\n\tv_mov_b32_e32 v0, 1\n\tv_mov_b32_e32 v48, 0\n\tglobal_atomic_add v48, v0, s[60:61]\n"
            "\ts_waitcnt vmcnt(0)\n.LVK_end:\n\ts_or_b64 exec, exec, s[68:69]\n")
    t = once(t, RELOAD_0, chk + RELOAD_0)
    t = once(t, "\t\t.amdhsa_next_free_vgpr 256\n\t\t.amdhsa_next_free_sgpr 96\n\t\t.amdhsa_accum_offset 256\n", "\t\t.amdhsa_next_free_vgpr 280\n\t\t.amdhsa_next_free_sgpr 96\n\t\t.amdhsa_accum_offset 256\n")
    v["verify_keys"] = t
    # the same check with the copies in LDS (12 KiB behind the kernel's own 18 KiB): occupancy stays at two wavefronts per SIMD
    t = once(base, PROLOGUE, PROLOGUE + "\ts_load_dwordx2 s[60:61], s[0:1], 0x38\n\ts_lshl_b32 s62, s2, 13\n")
    for anchor, quads in snap:
        first, rest = anchor.split("\n", 1)
        copies = "".join("\tds_write_b128 v192, v[%d:%d] offset:%d\n" % (r, r + 3, LDS_BYTES + (a // 4) * 2048) for a, r in quads)
        t = once(t, anchor, first + "\n" + copies + rest)
    chk = "\ts_waitcnt vmcnt(0)\n\ts_mov_b64 s[64:65], 0\n"
    for q, off in enumerate(offs):
        chk += ("\tv_lshl_add_u64 v[42:43], s[12:13], 0, v[192:193]\n\tv_add_co_u32_e32 v42, vcc, 0x%x, v42\n\ts_nop 1\n\tv_addc_co_u32_e32 v43, vcc, 0, v43, vcc\n"
                "\tglobal_load_dwordx4 v[44:47], v[42:43], off\n\tds_read_b64 v[48:49], v192 offset:%d\n\tds_read_b64 v[0:1], v192 offset:%d\n"
                "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n" % (off, LDS_BYTES + q * 2048, LDS_BYTES + q * 2048 + 8))
        chk += ("\tv_cmp_ne_u32_e32 vcc, v44, v48\n\ts_mov_b64 s[66:67], vcc\n\tv_cmp_ne_u32_e32 vcc, v45, v49\n\ts_or_b64 s[66:67], s[66:67], vcc\n"
                "\tv_cmp_ne_u32_e32 vcc, v46, v0\n\ts_or_b64 s[66:67], s[66:67], vcc\n\tv_cmp_ne_u32_e32 vcc, v47, v1\n\ts_or_b64 s[66:67], s[66:67], vcc\n"
                "\ts_or_b64 s[64:65], s[64:65], s[66:67]\n\ts_and_saveexec_b64 s[68:69], s[66:67]\n\ts_cbranch_execz .LVL_%d\n"
                "\tv_lshl_add_u32 v42, v192, 2, s62\n\tglobal_store_dwordx2 v42, v[48:49], s[60:61] offset:64\n\tglobal_store_dwordx2 v42, v[0:1], s[60:61] offset:72\n"
                "\tglobal_store_dwordx4 v42, v[44:47], s[60:61] offset:80\n\tv_mov_b32_e32 v43, %d\n\tglobal_store_dword v42, v43, s[60:61] offset:96\n"
                "\tv_mov_b32_e32 v43, s2\n\tglobal_store_dword v42, v43, s[60:61] offset:100\n\ts_waitcnt vmcnt(0)\n.LVL_%d:\n\ts_or_b64 exec, exec, s[68:69]\n" % (q, q + 1, q))
    chk += ("\ts_and_saveexec_b64 s[68:69], s[64:65]\n\ts_cbranch_execz .LVL_end\n\tv_mov_b32_e32 v0, 1\n\tv_mov_b32_e32 v48, 0\n\tglobal_atomic_add v48, v0, s[60:61]\n"
            "\ts_waitcnt vmcnt(0)\n.LVL_end:\n\ts_or_b64 exec, exec, s[68:69]\n")
    t = once(t, RELOAD_0, chk + RELOAD_0)
    v["verify_keys_lds"] = lds_size(t, LDS_BYTES + 6 * 2048)
    # control: the LDS copies alone, no check at the latch (is the failure still there with the copies in place?)
    t = base
    for anchor, quads in snap:
        first, rest = anchor.split("\n", 1)
        copies = "".join("\tds_write_b128 v192, v[%d:%d] offset:%d\n" % (r, r + 3, LDS_BYTES + (a // 4) * 2048) for a, r in quads)
        t = once(t, anchor, first + "\n" + copies + rest)
    v["copies_only_lds"] = lds_size(t, LDS_BYTES + 6 * 2048)
    # ---- sixth round: why is the first workgroup of a CU (blocks below 256) never wrong?  Its LDS base is 0 / it starts first / it has been alone for a while ----
    def shift_lds(text, by):
        out = []
        for x in text.split("\n"):
            if re.match(r"\tds_(?:read|write)", x):
                code, sep, com = x.partition(";")
                m = re.search(r" offset:(\d+)", code)
                code = (code[:m.start()] + " offset:%d" % (int(m.group(1)) + by) + code[m.end():]) if m else code.rstrip() + " offset:%d" % by
                x = code + ((" " + sep + com) if sep else "")
            out.append(x)
        return "\n".join(out)
    v["lds_shifted_20k"] = lds_size(shift_lds(base, 20480), 40960)
    v["first_blocks_exit"] = once(base, PROLOGUE, "; %bb.0:\n\ts_cmp_lt_u32 s2, 256\n\ts_cbranch_scc1 .LBB1_9\n\ts_load_dword s44, s[0:1], 0x24\n")
    if only:
        v = {k: t for k, t in v.items() if k in only}
    return v


def main():
    os.makedirs(WORK, exist_ok=True)
    base_s = os.path.join(WORK, "ep_fail.s")
    sh(["hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", base_s, SRC])
    base = open(base_s).read()
    assert ".private_segment_fixed_size: 28" in base, "this compiler no longer spills the failing build the same way: re-derive the anchors"
    only = set(sys.argv[1].split(',')) if len(sys.argv) > 1 else None
    for name, text in variants(base, only).items():
        s = os.path.join(WORK, name + ".s")
        open(s, "w").write(text)
        o, hsaco, fb = s[:-2] + ".o", s[:-2] + ".hsaco", s[:-2] + ".hipfb"
        sh([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
        sh([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", o, "-o", hsaco])
        sh([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null",
            "-input=" + hsaco, "-output=" + fb])
        sh(["hipcc"] + FLAGS + ["-fPIC", "-shared", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, SRC, "-o", os.path.join(OUT, "ep_sh_%s.so" % name)])
        print("built ep_sh_%s.so" % name)
    if only:
        return
    # the compiler's own build of the same source (not through the assembler), and the plain loop as the reference of what is right
    sh(["hipcc"] + FLAGS + ["-fPIC", "-shared", SRC, "-o", os.path.join(OUT, "ep_sh_hipcc.so")])
    sh(["hipcc"] + [f for f in FLAGS if "AB_FORM" not in f] + ["-DAB_FORM=1"] + ["-fPIC", "-shared", SRC, "-o", os.path.join(OUT, "ep_sh_plain.so")])


if __name__ == "__main__":
    main()
