"""Seeded inputs of tests/golden/callers.npz, shared by the generator (make_callers_golden.py, which adds the REFERENCE's outputs) and by
tests/test_oracle_golden.py (which recomputes the oracle's outputs from the same stream).  Oracle-side only: nothing here touches the reference."""
import numpy as np

N, L, BG, SIGMA = 1024, 2, 8, 2.98e-8          # SET_1 ring and gadget (test/benchmark.c:53-54)


def fdfb_multivalue(O):
    """full_domain_functional_bootstrap (src/bootstrap.c:519-538) and multivalue_bootstrap_CLOT21 (:222-230), 12-word LWE key."""
    rng = O.Rng(0xCA11E5)
    n, t, bb = 12, 5, 2
    lwe_s = O.gen_binary_key(rng, n)
    rlwe_s = O.gen_binary_key(rng, N).reshape(1, N)
    bk = O.gen_bootstrap_key(rng, lwe_s, rlwe_s, L, BG, SIGMA)
    ksk = O.gen_tlwe_ks_key(rng, rlwe_s.reshape(-1).copy(), lwe_s, t, bb, 2.0 ** -30)
    lut = O.u64(rng.words(8))
    tv = O.trlwe_torus_packing_many_LUT(lut, 1, N, 4, 2)
    cts = np.stack([O.tlwe_sample(rng, (m << 61) % 2 ** 64, lwe_s, 2.0 ** -30) for m in range(8)])
    lut16 = O.u64(rng.words(16))
    tv16 = O.trlwe_torus_packing(lut16, 1, N)
    c_mv = O.tlwe_sample(rng, O.double2torus(0.25), lwe_s, 2.0 ** -30)
    return dict(n=n, t=t, bb=bb, lwe_s=lwe_s, rlwe_s=rlwe_s, bk=bk, ksk=ksk, lut=lut, tv=tv, cts=cts, lut16=lut16, tv16=tv16, c_mv=c_mv)


def galois(O):
    """trlwe_keyswitch (src/keyswitch.c:162-193), trlwe_eval_automorphism (src/trlwe.c:775-781), functional_bootstrap_ga (src/bootstrap_ga.c:62-76)."""
    rng = O.Rng(0xCA11E6)
    n = 10
    s = O.gen_binary_key(rng, N)
    s2 = O.gen_binary_key(rng, N)
    ks = O.gen_trlwe_ks_key(rng, s2, s, 4, 8, SIGMA)
    c_ks = O.trlwe_sample(rng, O.u64(rng.words(N)), s2.reshape(1, N), SIGMA)
    ak = O.gen_automorphism_keyset(rng, s, L, BG, SIGMA)
    c_aut = O.trlwe_sample(rng, O.u64(rng.words(N)), s.reshape(1, N), SIGMA)
    gens = [1, 3, 2 * N - 1, 777]
    lwe_s = O.gen_binary_key(rng, n)
    bk = O.gen_bootstrap_key_ga(rng, lwe_s, s.reshape(1, N), L, BG, SIGMA)
    lut = O.u64(rng.words(4))
    tv = O.trlwe_torus_packing(lut, 1, N)
    cts = np.stack([O.tlwe_sample(rng, O.double2torus(m / 8.0), lwe_s, 1e-6) for m in range(4)])
    return dict(n=n, s=s, s2=s2, ks=ks, c_ks=c_ks, ak=ak, c_aut=c_aut, gens=gens, lwe_s=lwe_s, bk=bk, lut=lut, tv=tv, cts=cts)


def key_switches(O):
    """trlwe_priv_keyswitch_2 (src/keyswitch.c:52-63) and trlwe_packing1_keyswitch (:458-475)."""
    rng = O.Rng(0xCA11E7)
    sigma = 2.0 ** -40
    s = O.gen_binary_key(rng, N)
    ks0, ks1 = O.gen_priv_ks_key(rng, s, s, 10, 3, sigma)
    msg = np.zeros(N, dtype=np.uint64)
    msg[0] = O.double2torus(0.125)
    ct = O.trlwe_sample(rng, msg, s.reshape(1, N), sigma)
    s_in = O.gen_binary_key(rng, 48)
    kskb = O.gen_packing1_ks_key(rng, s_in, s, 5, 3, sigma)
    cs = np.stack([O.tlwe_sample(rng, O.double2torus(m), s_in, sigma) for m in (0.125, -0.25)])
    return dict(s=s, ks0=ks0, ks1=ks1, msg=msg, ct=ct, s_in=s_in, kskb=kskb, cs=cs)


def unfolded(O, unfolding=2):
    """functional_bootstrap with key->unfolding = 2 (src/bootstrap.c:23-48,124-149)."""
    rng = O.Rng(0xCA11E8)
    n, sigma = 24, 2.0 ** -40
    lwe_s = O.gen_binary_key(rng, n)
    s = O.gen_binary_key(rng, N)
    su = O.gen_bootstrap_key_unfolded(rng, lwe_s, s, L, BG, sigma, unfolding)
    lut = O.u64(rng.words(4))
    tv = O.trlwe_torus_packing(lut, 1, N)
    cts = np.stack([O.tlwe_sample(rng, O.double2torus(m / 8.0), lwe_s, 2.0 ** -20) for m in range(4)])
    return dict(n=n, lwe_s=lwe_s, s=s, su=su, lut=lut, tv=tv, cts=cts, unfolding=unfolding)
