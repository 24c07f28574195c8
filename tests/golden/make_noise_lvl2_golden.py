"""Golden fixture: the REFERENCE's own noise at N = 2048 (TFHEpp lvl2: n = 632, l = 4, Bg = 2^9) for the three batch criteria the GPU tests used to
derive by hand:

  pbs   |phase - LUT slot| of programmable_bootstrap (precision 3; test/tests.c:1545-1560's shape) over 2048 samples, both reference builds;
  ga    the same for functional_bootstrap_ga at n = 632 (test/tests.c:1614-1640's shape: torus_base 4), 512 samples -- blind_rotate_ga forces every
        mod-switched mask word odd (src/bootstrap_ga.c:44), so the error has a drift component on top of the noise;
  cb    circuit_bootstrap_3 at BASELINE.json configs[3]'s keys (test/tests.c:967-1003: packing key t = 6, base 2^4 -- 184,320 seed-compressed rows made by
        the reference's own trlwe_new_packing1_KS_key --, private pair t = 20, base 2^2): every output TRGSW multiplies a random TRLWE sample
        (trgsw_mul_trlwe_DFT) and the phase error of the product is taken per coefficient, 192 inputs x 2048 coefficients.  The error of one output is
        dominated by ONE rounding term per gadget level (the 40 dropped bits of the packing key switch land on X^0 and are spread by the private key
        switch's multiplication by the key), so its size varies from output to output like a chi-square with 4 degrees of freedom: stored are the rms,
        maximum and share within 2^58 of EVERY output, and the full error vectors of the first 24.  The amplification of that term depends on the
        random sample being multiplied and on the key (the negacyclic product of the sample's digit polynomials with the 0/1 key is dominated by a few
        low-frequency bins: +-20 % in rms between two samples), so the sample and its message are stored too and the GPU test uses the same keys
        (same seed) and the same sample.

Keys and ciphertexts come from the host layer's seeded generator (the same seeds as the GPU tests); everything is computed by oracle/_ref, the reference
compiled from /root/reference.  Stored: the error magnitudes (float32 log2 for the big one), seeds and parameters -- data only.

    python tests/golden/make_noise_lvl2_golden.py        (build container: needs oracle/_ref; ~10 GB of memory, a few minutes)
"""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SEED = 0x4E4F4953   # "NOIS"
B_PBS, B_GA, B_CB, B_CB_FULL = 2048, 512, 192, 24


def dist(ph, want):
    return np.abs((ph - want).astype(np.int64).astype(np.float64))


def stats(name, v):
    print("%-14s n %6d  max 2^%.2f  rms 2^%.2f  within 2^58: %.4f  within 2^57: %.4f" % (name, v.size, np.log2(v.max()), np.log2(np.sqrt((v ** 2).mean())),
                                                                                      (v < 2.0 ** 58).mean(), (v < 2.0 ** 57).mean()), flush=True)


def main():
    import mosfhet_amd as ma
    from mosfhet_amd import host
    from oracle import reflib
    reflib.build()
    P = dict(ma.PARAMS_LVL2)
    N, l, Bg, n = P["N"], P["l"], P["Bg_bit"], P["n"]
    host.seed(SEED)
    lk = host.LweKey(n, P["lwe_sigma"])
    rk = host.RlweKey(N, 1, P["rlwe_sigma"])
    s = rk.s[0]
    s_out = rk.extracted_lwe_key().s
    bk = host.gen_bootstrap_key(rk, lk, l, Bg)
    lut = np.array([host.double2torus(x) for x in (0.0625, 0.3125, -0.1875, 0.4375)], dtype=np.uint64)
    tv = host.torus_packing(lut, 1, N)
    res = {}

    # ---- programmable bootstrap
    cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B_PBS)], lk)
    want = lut[np.arange(B_PBS) % 4]
    for backend in ("avx512", "ffnt"):
        ref = reflib.get(backend)
        ref.init(N)
        h = ref.bk_new(bk, 1, l, Bg)
        with ThreadPoolExecutor(8) as pool:
            outs = list(pool.map(lambda c: ref.programmable_bootstrap(tv, c, h, 3, 0, 0), cts))
        ref.bk_free(h)
        res["pbs_" + backend] = dist(host.tlwe_phase(np.stack(outs), s_out), want)
        stats("pbs " + backend, res["pbs_" + backend])

    ref = reflib.get("avx512")
    ref.init(N)

    # ---- Galois-automorphism bootstrap at n = 632
    bk_ga = host.gen_bootstrap_key_ga(rk, lk, l, Bg)
    ak = host.gen_automorphism_keyset(rk, l, Bg)
    hg = ref.bk_ga_new(bk_ga, ak, l, Bg)
    cts_ga = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B_GA)], lk)
    with ThreadPoolExecutor(8) as pool:
        outs = list(pool.map(lambda c: ref.functional_bootstrap_ga(tv, c, hg, 4), cts_ga))
    ref.bk_ga_free(hg)
    del bk_ga, ak
    res["ga"] = dist(host.tlwe_phase(np.stack(outs), s_out), lut[np.arange(B_GA) % 4])
    stats("ga", res["ga"])

    # ---- circuit_bootstrap_3 with the reference's own packing key
    t0 = time.time()
    ref.l.ref_generic_key_new.restype = C.c_void_p
    s_c = np.ascontiguousarray(s)
    pk = C.c_void_p(ref.l.ref_generic_key_new(0, s_c.ctypes.data_as(C.c_void_p), N, s_c.ctypes.data_as(C.c_void_p), N, 6, 4, C.c_double(P["rlwe_sigma"])))
    print("reference packing key (184,320 rows): %.1f s" % (time.time() - t0), flush=True)
    kska = host.gen_priv_ks_key(rk, rk, 20, 2)
    hb = ref.bk_new(bk, 1, l, Bg)
    ms = [0.25 if b % 3 else 0.0 for b in range(B_CB)]
    cts_cb = host.tlwe_samples([host.double2torus(m) for m in ms], lk)
    rng = np.random.default_rng(SEED)
    msg = rng.integers(0, 2 ** 64, size=N, dtype=np.uint64)
    from oracle import oracle as O
    O.build()
    rnd = O.trlwe_sample(O.Rng(5), msg, s.reshape(1, N), P["rlwe_sigma"])
    errs = np.empty((B_CB, N))
    t0 = time.time()
    for b in range(B_CB):
        g = ref.circuit_bootstrap(cts_cb[b], hb, l, N, None, pk, 2, kska_flat=kska, bba=2)
        prod = ref.external_product(rnd, g, l, Bg)
        errs[b] = dist(O.trlwe_phase(prod, s.reshape(1, N)), msg if ms[b] else np.zeros(N, dtype=np.uint64))
    print("%d circuit bootstraps: %.1f s" % (B_CB, time.time() - t0), flush=True)
    ref.generic_key_free(pk)
    ref.bk_free(hb)
    stats("cb products", errs)
    stats("  selector 1", errs[[b for b in range(B_CB) if ms[b]]])
    stats("  selector 0", errs[[b for b in range(B_CB) if not ms[b]]])
    res["cb_log2"] = np.log2(errs[:B_CB_FULL] + 1.0).astype(np.float32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "noise_lvl2.npz"), seed=np.uint64(SEED), lut=lut,
                        pbs_avx512=res["pbs_avx512"], pbs_ffnt=res["pbs_ffnt"], ga=res["ga"], cb_log2=res["cb_log2"],
                        cb_rms=np.sqrt((errs ** 2).mean(axis=1)), cb_max=errs.max(axis=1), cb_within58=(errs < 2.0 ** 58).mean(axis=1),
                        cb_rnd=rnd, cb_msg=msg, cb_messages=np.array(ms), params=np.array([n, N, l, Bg, 6, 4, 20, 2], dtype=np.int64))


if __name__ == "__main__":
    main()
