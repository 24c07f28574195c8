"""Golden fixture: the REFERENCE's own phase-error distribution over a batch of SET_1 programmable bootstraps.

The reference asserts 2^58 on single samples (test/tests.c:1560); over thousands of samples the noise tail brushes that bound, so the batch checks of
this repository (bench.py, the full-size parity tests) use "all within 2^60 and at least 99.5 % within 2^58".  This script pins that criterion to what
the reference itself produces: 2048 programmable bootstraps (precision 3, kappa = theta = 0, 4-slot LUT: the shape of test/tests.c:1545-1560 and of
BASELINE.json configs[1]) with keys and ciphertexts from the host layer's seeded generator, run through BOTH reference builds (AVX-512 SPQLIOS and
portable FFNT, oracle/_ref built from /root/reference by oracle/ref/Makefile) and through the oracle.  Stored: per-sample |phase - LUT slot| for the
three, the seed and the parameters -- data only.

    python tests/golden/make_phase_error_golden.py        (build container: needs oracle/_ref)
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SEED, B = 0x50484153, 2048   # "PHAS"


def inputs():
    import mosfhet_amd as ma
    from mosfhet_amd import host
    P = dict(ma.PARAMS_SET1)
    host.seed(SEED)
    lk = host.LweKey(P["n"], P["lwe_sigma"])
    rk = host.RlweKey(P["N"], P["k"], P["rlwe_sigma"])
    bk = host.gen_bootstrap_key(rk, lk, P["l"], P["Bg_bit"])
    lut = np.array([host.double2torus(x) for x in (0.0625, 0.3125, -0.1875, 0.4375)], dtype=np.uint64)
    tv = host.torus_packing(lut, P["k"], P["N"])
    cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk)
    return P, lk, rk, bk, lut, tv, cts


def dist(ph, want):
    return np.abs((ph - want).astype(np.int64).astype(np.float64))


def main():
    from mosfhet_amd import host
    from oracle import oracle as O, reflib
    reflib.build()
    P, lk, rk, bk, lut, tv, cts = inputs()
    s_out = rk.extracted_lwe_key().s
    want = lut[np.arange(B) % 4]
    res = {}
    for backend in ("avx512", "ffnt"):
        ref = reflib.get(backend)
        ref.init(P["N"])
        h = ref.bk_new(bk, P["k"], P["l"], P["Bg_bit"])
        with ThreadPoolExecutor(8) as pool:
            outs = list(pool.map(lambda c: ref.programmable_bootstrap(tv, c, h, 3, 0, 0), cts))
        ref.bk_free(h)
        res[backend] = dist(host.tlwe_phase(np.stack(outs), s_out), want)
    bk_dft = O.bk_to_dft(bk, P["k"], P["l"])
    with ThreadPoolExecutor(8) as pool:
        outs = list(pool.map(lambda c: O.programmable_bootstrap(tv, c, bk_dft, P["l"], P["Bg_bit"], 3, 0, 0), cts))
    res["oracle"] = dist(host.tlwe_phase(np.stack(outs), s_out), want)
    for k, v in res.items():
        print("%-7s max 2^%.2f  within 2^58: %.4f  within 2^57: %.4f  rms 2^%.2f" % (k, np.log2(v.max()), (v < 2.0 ** 58).mean(), (v < 2.0 ** 57).mean(),
                                                                                np.log2(np.sqrt((v ** 2).mean()))))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "phase_error.npz"), seed=np.uint64(SEED), batch=np.int64(B),
                        err_avx512=res["avx512"], err_ffnt=res["ffnt"], err_oracle=res["oracle"], lut=lut, first_ct=cts[0], tv=tv)


if __name__ == "__main__":
    main()
