#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (antoniocgj/MOSFHET).

Run in the build container only (needs /root/reference): `python tests/golden/make_golden.py`.
It builds oracle/_ref (oracle/ref/Makefile: the reference's sources compiled where they lie, plus our flat-buffer
shim) and records INPUTS and the reference's OUTPUTS -- data only, no reference source -- so that the
`-m "not gpu"` tests can pin the oracle on machines where /root/reference does not exist.

Inputs come from the oracle's splitmix64 generator with fixed seeds (the reference's RNG is RDRAND-seeded and
not reproducible, src/misc.c:34-49).  Big keys are NOT stored: fixtures carry the seed, and the tests regenerate
the key with the same generator.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from oracle import reflib  # noqa: E402

S1 = dict(n=585, N=1024, k=1, l=2, Bg_bit=8, t=5, base_bit=2, lwe_sigma=9.141776004202573e-5, rlwe_sigma=2.989040792967434e-8)
L2 = dict(n=632, N=2048, k=1, l=4, Bg_bit=9, t=8, base_bit=4, lwe_sigma=2.0 ** -15, rlwe_sigma=2.0 ** -44)
ROT_AMOUNTS = lambda N: [0, 1, N - 1, N, N + 1, 2 * N - 1, 777 % (2 * N), 2 * N + 3, 5 * N + 1]  # noqa: E731


def edge_poly(rng, N, Bg_bit, l):
    """Uniform words plus the edge values SURVEY.md 8(c) lists: 0, 2^63, 2^64-1 and digit-boundary +-1."""
    p = O.u64(rng.words(N))
    p[0], p[1], p[2] = 0, 2 ** 63, 2 ** 64 - 1
    for j in range(l):
        b = (1 << (64 - (j + 1) * Bg_bit)) >> 1
        p[3 + 3 * j], p[4 + 3 * j], p[5 + 3 * j] = b % 2 ** 64, (b - 1) % 2 ** 64, (b + 1) % 2 ** 64
    return p


def integer_vectors(ref):
    out = {}
    rng = O.Rng(0x1001)
    for name, P in (("s1", S1), ("l2", L2)):
        N, Bg, l = P["N"], P["Bg_bit"], P["l"]
        p = edge_poly(rng, N, Bg, l)
        out["%s_poly" % name] = p
        out["%s_decompose_i" % name] = np.stack([ref.poly_decompose_i(p, Bg, l, i) for i in range(l)])
        out["%s_decompose" % name] = ref.poly_decompose(p, Bg, l)
        acc = O.u64(rng.words(N))
        out["%s_acc" % name] = acc
        amounts = ROT_AMOUNTS(N)
        out["%s_rot_amounts" % name] = np.array(amounts, dtype=np.int64)
        out["%s_mul_by_xai" % name] = np.stack([ref.poly_mul_by_xai(p, a, 0) for a in amounts])
        out["%s_mul_by_xai_minus_1" % name] = np.stack([ref.poly_mul_by_xai(p, a, 2) for a in amounts])
        out["%s_mul_by_xai_addto" % name] = np.stack([ref.poly_mul_by_xai(p, a, 1, acc=acc) for a in amounts])
        gens = [1, 3, 5, 2 * N - 1, N + 1, 4097 % (2 * N) | 1]
        out["%s_gens" % name] = np.array(gens, dtype=np.uint64)
        out["%s_permute" % name] = np.stack([ref.poly_permute(p, g) for g in gens])
        c = O.u64(rng.words(2 * N)).reshape(2, N)
        out["%s_trlwe" % name] = c
        idxs = [0, N // (2 * l), N - 1]
        out["%s_extract_idx" % name] = np.array(idxs, dtype=np.int64)
        out["%s_extract" % name] = np.stack([ref.trlwe_extract_tlwe(c, i) for i in idxs])
        lut = O.u64(rng.words(4))
        out["%s_lut" % name] = lut
        out["%s_packing" % name] = ref.trlwe_torus_packing(lut, 1, N)
    xs = O.u64(rng.words(64))
    xs[:4] = [0, 2 ** 63, 2 ** 64 - 1, 2 ** 52]
    out["torus2int_x"] = xs
    for ls in (10, 11, 12):
        out["torus2int_%d" % ls] = np.array([ref.torus2int(x, ls) for x in xs], dtype=np.uint64)
    ds = np.array([0.0, 0.125, -0.125, 1.0 / 3, 0.49, -0.49, 1.0 / 16, 1.0 / 32], dtype=np.float64)
    out["double2torus_x"] = ds
    out["double2torus"] = np.array([ref.double2torus(float(x)) for x in ds], dtype=np.uint64)
    return out


def keyswitch_vectors(ref):
    """tlwe_keyswitch at a toy size (n_in=64, n_out=16, t=2, bb=2) and with the t / base_bit of SET_1 and lvl2 at small n;
    the key is part of the fixture (small)."""
    out = {}
    rng = O.Rng(0x2002)
    for name, (n_in, n_out, t, bb) in (("toy", (64, 16, 2, 2)), ("s1like", (96, 24, 5, 2)), ("l2like", (32, 12, 8, 4))):
        s_in, s_out = O.gen_binary_key(rng, n_in), O.gen_binary_key(rng, n_out)
        ksk = O.gen_tlwe_ks_key(rng, s_in, s_out, t, bb, 2.0 ** -40)
        cts = np.stack([O.tlwe_sample(rng, O.double2torus(m / 8.0), s_in, 2.0 ** -40) for m in range(6)])
        cts[4, :-1] = 0
        cts[5, :-1] = 2 ** 64 - 1
        h = ref.ksk_new(ksk, bb)
        res = np.stack([ref.tlwe_keyswitch(c, h, n_out) for c in cts])
        ref.ksk_free(h)
        out.update({"%s_params" % name: np.array([n_in, n_out, t, bb], dtype=np.int64), "%s_s_in" % name: s_in,
                    "%s_s_out" % name: s_out, "%s_ksk" % name: ksk, "%s_in" % name: cts, "%s_out" % name: res})
    return out


def fft_vectors(refs):
    """Negacyclic products (64-bit x 10-bit, as test_poly_DFT_mul test/tests.c:244-276) with the exact naive product
    and BOTH reference back-ends' FFT results; plus DFT round trips."""
    out = {}
    rng = O.Rng(0x3003)
    for N in (1024, 2048):
        a = O.u64(rng.words(N))
        b = (O.u64(rng.words(N)) % np.uint64(1024)) - np.uint64(512)
        out["n%d_a" % N], out["n%d_b" % N] = a, b
        out["n%d_exact" % N] = refs["ffnt"].poly_naive_mul(a, b)
        for be, ref in refs.items():
            ref.init(N)
            out["n%d_prod_%s" % (N, be)] = ref.poly_mul_fft(a, b)
            out["n%d_roundtrip_%s" % (N, be)] = ref.poly_dft_roundtrip(a)
    return out


def external_product_vectors(refs):
    """One TRGSW (.) TRLWE at SET_1 and at lvl2 parameters: inputs in the torus domain, outputs of both back-ends."""
    out = {}
    rng = O.Rng(0x4004)
    for name, P in (("s1", S1), ("l2", L2)):
        N, l, Bg = P["N"], P["l"], P["Bg_bit"]
        s = O.gen_binary_key(rng, N).reshape(1, N)
        msg = O.u64(rng.words(N))
        g = O.trgsw_monomial_sample(rng, 1, 5, s, l, Bg, P["rlwe_sigma"])
        c = O.trlwe_sample(rng, msg, s, P["rlwe_sigma"])
        out.update({"%s_key" % name: s, "%s_msg" % name: msg, "%s_trgsw" % name: g, "%s_trlwe" % name: c})
        for be, ref in refs.items():
            ref.init(N)
            out["%s_out_%s" % (name, be)] = ref.external_product(c, g, l, Bg)
    return out


def bootstrap_vectors(refs):
    """End-to-end programmable / functional bootstraps of the reference at SET_1 (N=1024).  The bootstrap key is
    regenerated from `seed` by the tests (oracle generator), so only seeds, inputs and reference outputs are stored.
    Cases: full n=585 (phase-level comparison) and a short n=8 key (few CMUX steps: ciphertexts stay close)."""
    out = {}
    P = S1
    for name, n, seed in (("full", P["n"], 0x5005), ("short", 8, 0x6006)):
        rng = O.Rng(seed)
        lwe_s = O.gen_binary_key(rng, n)
        rlwe_s = O.gen_binary_key(rng, P["N"]).reshape(1, P["N"])
        bk = O.gen_bootstrap_key(rng, lwe_s, rlwe_s, P["l"], P["Bg_bit"], P["rlwe_sigma"])
        lut = O.u64(rng.words(4))
        tv = O.trlwe_torus_packing(lut, 1, P["N"])
        cts = np.stack([O.tlwe_sample(rng, O.double2torus(m / 8.0), lwe_s, P["lwe_sigma"]) for m in range(4)])
        ct_k = O.tlwe_sample(rng, 0xA << 58, lwe_s, P["lwe_sigma"])  # tests.c:1562: kappa = 3 case
        out.update({"%s_seed" % name: np.array([seed], dtype=np.uint64), "%s_n" % name: np.array([n]),
                    "%s_lut" % name: lut, "%s_cts" % name: cts, "%s_ct_kappa" % name: ct_k})
        for be, ref in refs.items():
            ref.init(P["N"])
            h = ref.bk_new(bk, 1, P["l"], P["Bg_bit"])
            out["%s_pbs_%s" % (name, be)] = np.stack([ref.programmable_bootstrap(tv, c, h, 3, 0, 0) for c in cts])
            out["%s_pbs_kappa_%s" % (name, be)] = ref.programmable_bootstrap(tv, ct_k, h, 3, 3, 0)
            out["%s_fb_%s" % (name, be)] = np.stack([ref.functional_bootstrap(tv, c, h, 4) for c in cts])
            out["%s_wo_extract_%s" % (name, be)] = ref.functional_bootstrap_wo_extract(tv, cts[1], h, 4)
            ref.bk_free(h)
    return out


def main():
    if not reflib.build():
        sys.exit("/root/reference is not present: golden vectors can only be generated in the build container")
    refs = {be: reflib.get(be) for be in ("avx512", "ffnt") if reflib.available(be)}
    assert "ffnt" in refs, "the portable reference build is required"
    np.savez_compressed(os.path.join(HERE, "integer_ops.npz"), **integer_vectors(refs["ffnt"]))
    np.savez_compressed(os.path.join(HERE, "keyswitch.npz"), **keyswitch_vectors(refs["ffnt"]))
    np.savez_compressed(os.path.join(HERE, "fft_products.npz"), **fft_vectors(refs))
    np.savez_compressed(os.path.join(HERE, "external_product.npz"), **external_product_vectors(refs))
    np.savez_compressed(os.path.join(HERE, "bootstrap.npz"), **bootstrap_vectors(refs))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-24s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
