#!/usr/bin/env python3
"""gen_ks_words_asm.py -- writes mosfhet_amd/csrc/ks_words_asm.inc: the consume loop of table_ks_words_kernel (keyswitch_words_kernels.h).

One inline-assembly block per candidate count (2^bb - 1 = 3, 7, 15) that, for the `npos` digit positions of one stage, adds to each of the wavefront's 64
accumulators (one ciphertext each, the lane's output word) the candidate word its digit selects:

    acc[c] += cand[pos][digit(c, pos)]          c < 64,  pos < npos

The digit of (ciphertext, position) is wave-uniform, so the candidate is selected by VGPR-relative addressing (S_SET_GPR_IDX mode, SRC0_REL: the
add's src0 is v[block + M0[7:0]]): per add ONE SALU (M0 <- the 16-bit table entry `0x1000 | 2 * digit`, two entries per SGPR) and ONE v_lshl_add_u64 --
no LDS gather, no per-lane select (tools/ubench/gen_gpridx.py measures the pair: 4.3 clocks per add and SIMD at two wavefronts per SIMD, against ~12
for the LDS-gather form of keyswitch_kernels.h).  Positions are software-pipelined inside the block: while position p is added, the entries (s_load, 128
bytes) and the candidate words (ds_read_b64 x cands) of position p + 1 are in flight into the other register set.

Register plan (fixed physical registers, declared as clobbers; the accumulators are ordinary "+v" operands):
    v[192:223]  candidate block X (even positions): word 0 = the zero row of digit 0, candidate v at 2 (v + 1)
    v[224:255]  candidate block Y (odd positions)
    s[36:67]    entries of an even position, s[68:99] of an odd one;  s[34:35] the running address of the entries
Operands: %0 .. %63 accumulators; %[la] LDS byte address of the lane's word in candidate row 0 of the stage (advanced inside); %[dp] address of the
wavefront's entries of the stage; %[np] positions left (counted down inside).
LDS-DMA of a later stage, spread over the block (one global_load_lds_dwordx4 behind the adds of a position, so that its issue costs the wavefront one slot
instead of a burst's queueing): %[nd] DMAs left to issue, %[ga] the lane's source address, %[gs] the (wave-uniform) byte step to the wavefront's next row
pair, %[ld] LDS byte address the DMA writes to (its step: 8 KiB = the workgroup's eight row pairs), %[na] times the address still advances (a wavefront
with fewer row pairs than DMAs repeats its last one: same bytes to the same place, and every wavefront's vmcnt counts the same).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "mosfhet_amd", "csrc", "ks_words_asm.inc")
X, Y, SA, SB = 192, 224, 36, 68
UNROLLED = ((15, 6), (15, 4))     # (candidates, positions per stage) that get a written-out block: the packing switch and the lvl2 LWE switch of the BASELINE configs


EXP = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else set()     # timing experiments: "nosload" (entries loaded for the first position only), "nods"


def loads(cands, blk, sset, pos_off, first=False):
    """entries and candidate words of the position `pos_off` positions after the one %[la] / %[dp] point at"""
    out = [] if ("nosload" in EXP and not first) else ["s_load_dwordx16 s[%d:%d], s[34:35], 0x%x" % (sset, sset + 15, 128 * pos_off),
           "s_load_dwordx16 s[%d:%d], s[34:35], 0x%x" % (sset + 16, sset + 31, 128 * pos_off + 64)]
    for v in range(0 if "nods" not in EXP or first else cands, cands):
        out.append("ds_read_b64 v[%d:%d], %%[la] offset:%d" % (blk + 2 * (v + 1), blk + 2 * (v + 1) + 1, (pos_off * cands + v) * 512))
    return out


def adds(blk, sset):
    out = ["s_set_gpr_idx_on s%d, 0x1" % sset]
    for c in range(64):
        s = sset + c // 2
        out.append("s_mov_b32 m0, s%d" % s if c % 2 == 0 else "s_lshr_b32 m0, s%d, 16" % s)
        out.append("v_lshl_add_u64 %%%d, v[%d:%d], 0, %%%d" % (c, blk, blk + 1, c))
    out.append("s_set_gpr_idx_off")
    return out


def dma(tag):
    return ["s_cmp_eq_u32 %[nd], 0", "s_cbranch_scc1 %sf" % tag,
            "s_mov_b32 m0, %[ld]", "s_sub_u32 %[nd], %[nd], 1", "global_load_lds_dwordx4 %[ga], off",      # (the s_sub is the wait state between the M0 write and the DMA)
            "s_cmp_eq_u32 %[na], 0", "s_cbranch_scc1 %sf" % tag,
            "v_lshl_add_u64 %[ga], %[ga], 0, %[gs]", "s_add_u32 %[ld], %[ld], 0x2000", "s_sub_u32 %[na], %[na], 1", "%s:" % tag]


def block(cands):
    L = []
    L += ["s_mov_b64 s[34:35], %[dp]"]
    L += ["v_mov_b32 v%d, 0" % r for r in (X, X + 1, Y, Y + 1)]
    L += loads(cands, X, SA, 0, True)
    if EXP:
        L += loads(cands, Y, SB, 1, True)
    L += ["1:", "s_waitcnt lgkmcnt(0)", "s_cmp_lt_u32 %[np], 2", "s_cbranch_scc1 2f"]
    L += loads(cands, Y, SB, 1)
    L += ["2:"]
    L += adds(X, SA)
    L += dma("5")
    L += ["s_cmp_lt_u32 %[np], 2", "s_cbranch_scc1 4f", "s_waitcnt lgkmcnt(0)", "s_cmp_lt_u32 %[np], 3", "s_cbranch_scc1 3f"]
    L += loads(cands, X, SA, 2)
    L += ["3:"]
    L += adds(Y, SB)
    L += dma("6")
    L += ["s_add_u32 s34, s34, 0x100", "s_addc_u32 s35, s35, 0", "v_add_u32 %%[la], %d, %%[la]" % (2 * cands * 512),
          "s_sub_u32 %[np], %[np], 2", "s_cmp_lg_u32 %[np], 0", "s_cbranch_scc1 1b", "4:"]
    return L


def block_unrolled(cands, jb):
    """the same block with its `jb` positions written out: no position counter, no pointer updates (immediate offsets), the DMA slots compare against constants"""
    L = ["s_mov_b64 s[34:35], %[dp]"]
    L += ["v_mov_b32 v%d, 0" % r for r in (X, X + 1, Y, Y + 1)]
    L += loads(cands, X, SA, 0)
    for p in range(jb):
        blk, sset = (X, SA) if p % 2 == 0 else (Y, SB)
        L += ["s_waitcnt lgkmcnt(0)"]
        if p + 1 < jb:
            L += loads(cands, Y if p % 2 == 0 else X, SB if p % 2 == 0 else SA, p + 1)
        L += adds(blk, sset)
        if p >= 1:
            L += ["s_cmp_lt_u32 %%[na], %d" % p, "s_cbranch_scc1 %df" % (10 + p), "v_lshl_add_u64 %[ga], %[ga], 0, %[gs]", "s_add_u32 %[ld], %[ld], 0x2000", "%d:" % (10 + p)]
        L += ["s_cmp_le_u32 %%[nd], %d" % p, "s_cbranch_scc1 %df" % (20 + p), "s_mov_b32 m0, %[ld]", "s_nop 0", "global_load_lds_dwordx4 %[ga], off", "%d:" % (20 + p)]
    return L


def main():
    with open(OUT, "w") as f:
        f.write("// ks_words_asm.inc -- GENERATED by tools/gen_ks_words_asm.py (see its header for the register plan); included by keyswitch_words_kernels.h\n")
        f.write("// clang-format off\n")
        for cands in (3, 7, 15):
            f.write("#define KS_WORDS_CONSUME_%d \\\n" % cands)
            lines = block(cands)
            text = ['  "%s\\n\\t"' % ln for ln in lines]
            f.write(" \\\n".join(text) + "\n\n")
        for cands, jb in UNROLLED:
            f.write("#define KS_WORDS_CONSUME_%d_U%d \\\n" % (cands, jb))
            f.write(" \\\n".join('  "%s\\n\\t"' % ln for ln in block_unrolled(cands, jb)) + "\n\n")
        f.write("// clang-format on\n")


if __name__ == "__main__":
    main()
