"""Live comparison of the oracle with the reference build in oracle/_ref (compiled from /root/reference by
oracle/ref/Makefile; the .so travels with the snapshot).  Skipped where that build product is absent.
Fresh random inputs every run (seeded from the test id), complementing the fixed golden vectors."""
import numpy as np
import pytest

from oracle import reflib

BACKENDS = [b for b in ("avx512", "ffnt") if reflib.available(b)]
pytestmark = pytest.mark.skipif(not BACKENDS, reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(params=BACKENDS)
def ref(request):
    r = reflib.get(request.param)
    r.init(1024)
    r.init(2048)
    return r


def test_integer_ops_random(oracle, ref):
    rng = oracle.Rng(0xABC)
    for N, Bg, l in ((1024, 8, 2), (2048, 9, 4), (1024, 23, 1), (512, 6, 3)):
        p, acc = oracle.u64(rng.words(N)), oracle.u64(rng.words(N))
        for i in range(l):
            assert (ref.poly_decompose_i(p, Bg, l, i) == oracle.poly_decompose_i(p, Bg, l, i)).all()
        assert (ref.poly_decompose(p, Bg, l) == oracle.poly_decompose(p, Bg, l)).all()
        for a in [0, 1, N - 1, N, N + 1, 2 * N - 1] + [int(x % (4 * N)) for x in rng.words(6)]:
            assert (ref.poly_mul_by_xai(p, a, 0) == oracle.poly_mul_by_xai(p, a)).all()
            assert (ref.poly_mul_by_xai(p, a, 2) == oracle.poly_mul_by_xai_minus_1(p, a)).all()
            assert (ref.poly_mul_by_xai(p, a, 1, acc=acc) == oracle.poly_mul_by_xai_addto(acc, p, a)).all()
        for gen in (1, 3, 2 * N - 1, int(rng.next() % (2 * N)) | 1):
            assert (ref.poly_permute(p, gen) == oracle.poly_permute(p, gen)).all()
        c = oracle.u64(rng.words(2 * N)).reshape(2, N)
        for idx in (0, N // 3, N - 1):
            assert (ref.trlwe_extract_tlwe(c, idx) == oracle.trlwe_extract_tlwe(c, idx)).all()
    for x in rng.words(32):
        for ls in (10, 11, 12):
            assert ref.torus2int(x, ls) == oracle.torus2int(x, ls)


def test_fft_and_external_product_random(oracle, ref):
    rng = oracle.Rng(0xDEF)
    for N, l, Bg, sigma in ((1024, 2, 8, 2.98e-8), (2048, 4, 9, 2.0 ** -44)):
        a = oracle.u64(rng.words(N))
        b = (oracle.u64(rng.words(N)) % np.uint64(1024)) - np.uint64(512)
        exact = oracle.poly_naive_mul(a, b)
        assert (ref.poly_naive_mul(a, b) == exact).all()
        assert oracle.torus_dist(oracle.poly_mul_fft(a, b), exact).max() < 2.0 ** 40
        assert oracle.torus_dist(oracle.poly_mul_fft(a, b), ref.poly_mul_fft(a, b)).max() < 2.0 ** 32
        s = oracle.gen_binary_key(rng, N).reshape(1, N)
        g = oracle.trgsw_monomial_sample(rng, 1, 7, s, l, Bg, sigma)
        c = oracle.trlwe_sample(rng, a, s, sigma)
        mine = oracle.external_product(c, oracle.trgsw_to_dft(g, 1, l), l, Bg)
        assert oracle.torus_dist(mine, ref.external_product(c, g, l, Bg)).max() < 2.0 ** 32


def test_by_component_product_order_is_held_to_the_reference_too(oracle, ref):
    """The oracle's second summation order (orc_set_product_order(1): the l rows of each accumulator component chained from zero, the partial sums added -- what a
    bootstrap split over one workgroup per component computes, mosfhet_amd/csrc/bootstrap_kernels.h: pbs_split_kernel) against the reference itself, at the SAME
    tolerances as the reference order: external products at both rings (2^32 on the ciphertext), a blind rotation and programmable bootstraps over a short key at
    SET_1's ring (2^38 on the ciphertext: no digit has diverged yet) and at the lvl2 gadget (by PHASE: with 36 bits of digits a rounding difference of 2^28 flips a
    last digit somewhere in every step, in this order as in the reference order -- the ciphertexts then differ by a key row and decrypt alike).  It differs from
    the reference order by no more than either differs from the reference."""
    rng = oracle.Rng(0xC0DE)
    for N, l, Bg, sigma in ((1024, 2, 8, 2.98e-8), (2048, 4, 9, 2.0 ** -44)):
        s = oracle.gen_binary_key(rng, N).reshape(1, N)
        g = oracle.trgsw_monomial_sample(rng, 1, 7, s, l, Bg, sigma)
        c = oracle.trlwe_sample(rng, oracle.u64(rng.words(N)), s, sigma)
        g_dft = oracle.trgsw_to_dft(g, 1, l)
        plain = oracle.external_product(c, g_dft, l, Bg)
        with oracle.product_order("by_component"):
            mine = oracle.external_product(c, g_dft, l, Bg)
        assert (oracle.external_product(c, g_dft, l, Bg) == plain).all()          # the switch is back
        assert oracle.torus_dist(mine, ref.external_product(c, g, l, Bg)).max() < 2.0 ** 32
        assert oracle.torus_dist(mine, plain).max() < 2.0 ** 30
    for N, l, Bg, sigma, by_phase in ((1024, 2, 8, 2.98e-8, False), (2048, 4, 9, 2.0 ** -44, True)):
        n = 12
        lwe_s = oracle.gen_binary_key(rng, n)
        rlwe_s = oracle.gen_binary_key(rng, N).reshape(1, N)
        bk = oracle.gen_bootstrap_key(rng, lwe_s, rlwe_s, l, Bg, sigma)
        bk_dft = oracle.bk_to_dft(bk, 1, l)
        h = ref.bk_new(bk, 1, l, Bg)
        lut = oracle.u64(rng.words(4))
        tv = oracle.trlwe_torus_packing(lut, 1, N)
        c = oracle.tlwe_sample(rng, oracle.double2torus(0.125), lwe_s, 2.0 ** -25)
        acc = oracle.trlwe_sample(rng, oracle.u64(rng.words(N)), rlwe_s, sigma)
        a = np.ascontiguousarray(c[:-1])
        out_s = rlwe_s.reshape(-1)

        def close(mine, theirs, trlwe):
            if not by_phase:
                return oracle.torus_dist(mine, theirs).max() < 2.0 ** 38
            ph = (lambda x: oracle.trlwe_phase(x, rlwe_s)) if trlwe else (lambda x: oracle.tlwe_phase(x, out_s))
            return np.max(oracle.torus_dist(ph(mine), ph(theirs))) < 2.0 ** 46

        plain = oracle.programmable_bootstrap(tv, c, bk_dft, l, Bg, 3, 0, 0)
        assert close(plain, ref.programmable_bootstrap(tv, c, h, 3, 0, 0), False)        # (the reference order under the same criterion)
        with oracle.product_order("by_component"):
            assert close(oracle.blind_rotate(acc, a, bk_dft, l, Bg), ref.blind_rotate(acc, a, h), True)
            got = oracle.programmable_bootstrap(tv, c, bk_dft, l, Bg, 3, 0, 0)
            assert close(got, ref.programmable_bootstrap(tv, c, h, 3, 0, 0), False)
            got2 = oracle.programmable_bootstrap(tv, c, bk_dft, l, Bg, 4, 2, 1)
            assert close(got2, ref.programmable_bootstrap(tv, c, h, 4, 2, 1), False)
        assert (got != plain).any() and close(got, plain, False)
        ref.bk_free(h)


def test_keyswitch_random(oracle, ref):
    rng = oracle.Rng(0x123)
    n_in, n_out, t, bb = 80, 20, 5, 2
    s_in, s_out = oracle.gen_binary_key(rng, n_in), oracle.gen_binary_key(rng, n_out)
    ksk = oracle.gen_tlwe_ks_key(rng, s_in, s_out, t, bb, 2.0 ** -40)
    h = ref.ksk_new(ksk, bb)
    for m in range(5):
        c = oracle.tlwe_sample(rng, oracle.double2torus(m / 8.0), s_in, 2.0 ** -40)
        assert (ref.tlwe_keyswitch(c, h, n_out) == oracle.tlwe_keyswitch(c, ksk, n_out, t, bb)).all()
    ref.ksk_free(h)


def test_blind_rotate_and_bootstrap_short_key(oracle, ref):
    """n = 12 CMUX steps at SET_1 ring parameters: ciphertext-level agreement (no divergence yet)."""
    rng = oracle.Rng(0x777)
    N, l, Bg, sigma, n = 1024, 2, 8, 2.98e-8, 12
    lwe_s = oracle.gen_binary_key(rng, n)
    rlwe_s = oracle.gen_binary_key(rng, N).reshape(1, N)
    bk = oracle.gen_bootstrap_key(rng, lwe_s, rlwe_s, l, Bg, sigma)
    bk_dft = oracle.bk_to_dft(bk, 1, l)
    h = ref.bk_new(bk, 1, l, Bg)
    tv = oracle.trlwe_torus_packing(oracle.u64(rng.words(4)), 1, N)
    c = oracle.tlwe_sample(rng, oracle.double2torus(0.125), lwe_s, 9.1e-5)
    acc = oracle.u64(rng.words(2 * N)).reshape(2, N)
    a = np.ascontiguousarray(c[:-1])
    assert oracle.torus_dist(oracle.blind_rotate(acc, a, bk_dft, l, Bg), ref.blind_rotate(acc, a, h)).max() < 2.0 ** 38
    got = oracle.programmable_bootstrap(tv, c, bk_dft, l, Bg, 3, 0, 0)
    assert oracle.torus_dist(got, ref.programmable_bootstrap(tv, c, h, 3, 0, 0)).max() < 2.0 ** 38
    got = oracle.programmable_bootstrap(tv, c, bk_dft, l, Bg, 4, 2, 1)
    assert oracle.torus_dist(got, ref.programmable_bootstrap(tv, c, h, 4, 2, 1)).max() < 2.0 ** 38
    ref.bk_free(h)


def test_fdfb_and_multivalue_short_key(oracle, ref):
    """full_domain_functional_bootstrap (src/bootstrap.c:519-538) and multivalue_bootstrap_CLOT21 (:222-230) with a
    12-bit LWE key at SET_1 ring parameters: the integer glue (sign LUT, b -= sign, key switch, add, many-LUT packing,
    extraction offsets) must agree with the reference; FFT parts within the short-key ciphertext tolerance."""
    rng = oracle.Rng(0x999)
    N, l, Bg, sigma, n, t, bb = 1024, 2, 8, 2.98e-8, 12, 5, 2
    lwe_s = oracle.gen_binary_key(rng, n)
    rlwe_s = oracle.gen_binary_key(rng, N).reshape(1, N)
    bk = oracle.gen_bootstrap_key(rng, lwe_s, rlwe_s, l, Bg, sigma)
    bk_dft = oracle.bk_to_dft(bk, 1, l)
    ksk = oracle.gen_tlwe_ks_key(rng, rlwe_s.reshape(-1).copy(), lwe_s, t, bb, 2.0 ** -30)
    h, kh = ref.bk_new(bk, 1, l, Bg), ref.ksk_new(ksk, bb)
    lut = oracle.u64(rng.words(8))
    tv = oracle.trlwe_torus_packing_many_LUT(lut, 1, N, 4, 2)
    assert (tv == ref.trlwe_torus_packing_many_LUT(lut, 1, N, 4, 2)).all()
    for m in range(8):
        c = oracle.tlwe_sample(rng, (m << 61) % 2 ** 64, lwe_s, 2.0 ** -30)
        mine = oracle.full_domain_functional_bootstrap(tv, c, bk_dft, ksk, l, Bg, t, bb, 3)
        theirs = ref.full_domain_functional_bootstrap(tv, c, h, kh, 3)
        assert oracle.torus_dist(mine, theirs).max() < 2.0 ** 40, m
    lut16 = oracle.u64(rng.words(16))
    tv16 = oracle.trlwe_torus_packing(lut16, 1, N)
    c = oracle.tlwe_sample(rng, oracle.double2torus(0.25), lwe_s, 2.0 ** -30)
    mine = oracle.multivalue_bootstrap_CLOT21(tv16, c, bk_dft, l, Bg, 2, 8)
    theirs = ref.multivalue_bootstrap_CLOT21(tv16, c, h, 2, 8)
    assert oracle.torus_dist(mine, theirs).max() < 2.0 ** 38
    ref.bk_free(h)
    ref.ksk_free(kh)


def test_trlwe_keyswitch_automorphism_and_ga_bootstrap(oracle, ref):
    """FFT-based trlwe_keyswitch (src/keyswitch.c:162-193), trlwe_eval_automorphism (src/trlwe.c:775-781), inverse_mod_2N
    (src/misc.c:142-159) and functional_bootstrap_ga (src/bootstrap_ga.c) with a 10-word LWE key."""
    rng = oracle.Rng(0x6A)
    N, l, Bg, sigma, n = 1024, 2, 8, 2.98e-8, 10
    for x in (1, 3, 5, 1023, 1025, 2047):
        assert ref.inverse_mod_2N(x, N) == oracle.inverse_mod_2N(x, N)
    s = oracle.gen_binary_key(rng, N)
    s2 = oracle.gen_binary_key(rng, N)
    ks = oracle.gen_trlwe_ks_key(rng, s2, s, 4, 8, sigma)
    c = oracle.trlwe_sample(rng, oracle.u64(rng.words(N)), s2.reshape(1, N), sigma)
    mine = oracle.trlwe_keyswitch(c, oracle.ks_to_dft(ks), 4, 8)
    assert oracle.torus_dist(mine, ref.trlwe_keyswitch(c, ks, 8)).max() < 2.0 ** 34
    ak = oracle.gen_automorphism_keyset(rng, s, l, Bg, sigma)
    ak_dft = oracle.ks_to_dft(ak)
    c = oracle.trlwe_sample(rng, oracle.u64(rng.words(N)), s.reshape(1, N), sigma)
    for gen in (1, 3, 2 * N - 1, 777):
        mine = oracle.trlwe_eval_automorphism(c, gen, ak_dft[(gen - 1) // 2], l, Bg)
        assert oracle.torus_dist(mine, ref.trlwe_eval_automorphism(c, gen, ak[(gen - 1) // 2], Bg)).max() < 2.0 ** 34, gen
    lwe_s = oracle.gen_binary_key(rng, n)
    bk = oracle.gen_bootstrap_key_ga(rng, lwe_s, s.reshape(1, N), l, Bg, sigma)
    bk_dft = oracle.bk_to_dft(bk, 1, l)
    h = ref.bk_ga_new(bk, ak, l, Bg)
    lut = oracle.u64(rng.words(4))
    tv = oracle.trlwe_torus_packing(lut, 1, N)
    for m in range(4):
        ct = oracle.tlwe_sample(rng, oracle.double2torus(m / 8.0), lwe_s, 1e-6)
        mine = oracle.functional_bootstrap_ga(tv, ct, bk_dft, ak_dft, l, Bg, 4)
        assert oracle.torus_dist(mine, ref.functional_bootstrap_ga(tv, ct, h, 4)).max() < 2.0 ** 42, m
        assert oracle.torus_dist(oracle.tlwe_phase(mine, s), lut[m]) < 2.0 ** 58
    ref.bk_ga_free(h)


def test_circuit_bootstrap_pieces(oracle, ref):
    """trlwe_priv_keyswitch_2 (src/keyswitch.c:52-63) against the reference, and trlwe_packing1_keyswitch
    (src/keyswitch.c:458-475): the reference is built with seed-compressed key rows, so its table-lookup loop is restated on
    plain rows in oracle/ref/ref_harness.c (same digit rule, the library's own trlwe_subto) -- bit-exact."""
    rng = oracle.Rng(0x7B)
    N, sigma = 1024, 2.0 ** -40
    s = oracle.gen_binary_key(rng, N)
    ks0, ks1 = oracle.gen_priv_ks_key(rng, s, s, 10, 3, sigma)
    msg = np.zeros(N, dtype=np.uint64)
    msg[0] = oracle.double2torus(0.125)
    ct = oracle.trlwe_sample(rng, msg, s.reshape(1, N), sigma)
    mine = oracle.trlwe_priv_keyswitch_2(ct, oracle.ks_to_dft(ks0), oracle.ks_to_dft(ks1), 10, 3)
    assert oracle.torus_dist(mine, ref.trlwe_priv_keyswitch_2(ct, ks0, ks1, 3)).max() < 2.0 ** 36
    want = oracle.poly_naive_mul(np.uint64(0) - s, msg)     # TRLWE(m) -> TRLWE(-s m)
    assert oracle.torus_dist(oracle.trlwe_phase(mine, s.reshape(1, N)), want).max() < 2.0 ** 52  # test_trlwe_pack_key_priv_ks tolerance
    n_in = 48
    s_in = oracle.gen_binary_key(rng, n_in)
    kskb = oracle.gen_packing1_ks_key(rng, s_in, s, 5, 3, sigma)
    for m in (0.125, -0.25):
        c = oracle.tlwe_sample(rng, oracle.double2torus(m), s_in, sigma)
        mine = oracle.trlwe_packing1_keyswitch(c, kskb, 3)
        assert (mine == ref.trlwe_packing1_keyswitch(c, kskb, 3)).all()
        ph = oracle.trlwe_phase(mine, s.reshape(1, N))
        assert oracle.torus_dist(ph[0], oracle.double2torus(m)) < 2.0 ** 58 and oracle.torus_dist(ph[1:], np.zeros(N - 1, dtype=np.uint64)).max() < 2.0 ** 58


# ---------------- callers either side of the bootstrap (oracle_ext.c) ----------------
@pytest.fixture(scope="module")
def wide(oracle):
    """Key material shared by the wider-path tests: N = 1024 ring, 40-word LWE key, gadget l = 3 / Bg = 2^10 at the 2^-44
    noise level; packing and private table-lookup keys are produced by the REFERENCE's key generation (per backend)."""
    rng = oracle.Rng(0x51DE)
    N, n, l, Bg, sigma = 1024, 40, 3, 10, 2.0 ** -44
    lwe_s = oracle.gen_binary_key(rng, n)
    s = oracle.gen_binary_key(rng, N)
    bk = oracle.gen_bootstrap_key(rng, lwe_s, s.reshape(1, N), l, Bg, sigma)
    return dict(rng=rng, N=N, n=n, l=l, Bg=Bg, sigma=sigma, lwe_sigma=2.0 ** -22, lwe_s=lwe_s, s=s, bk=bk, bk_dft=oracle.bk_to_dft(bk, 1, l), per_ref={})


def _ref_keys(wide, ref):
    """Reference-made table-lookup keys (packing t=12 bb=2, private t=4 bb=3), exported to flat rows for the oracle."""
    K = wide["per_ref"].get(ref.backend)
    if K is None:
        bkh = ref.bk_new(wide["bk"], 1, wide["l"], wide["Bg"])
        pkh, pk = ref.generic_key_new(0, wide["s"], wide["s"], 12, 2, wide["sigma"])
        skh, sk = ref.generic_key_new(1, wide["s"], wide["s"], 4, 3, wide["sigma"])
        K = wide["per_ref"][ref.backend] = dict(bkh=bkh, pkh=pkh, pk=pk, skh=skh, sk=sk)
    return K


def test_reference_made_table_keys_and_their_key_switches(oracle, ref, wide):
    """trlwe_packing1_keyswitch / trlwe_priv_keyswitch (src/keyswitch.c:458-475,639-656) -- the LIBRARY's loops on the library's
    own (seed-compressed) keys vs the oracle on the exported rows: integer work, bit-exact.  Also checks the exported rows
    decrypt to what the oracle's key generators encrypt."""
    K, N, s, rng = _ref_keys(wide, ref), wide["N"], wide["s"], wide["rng"]
    assert K["pk"].shape == (N, 12, 3, 2, N) and K["sk"].shape == (N + 1, 4, 7, 2, N)
    for i, j, v in ((0, 0, 1), (5, 3, 3), (N - 1, 11, 2)):
        ph = oracle.trlwe_phase(np.ascontiguousarray(K["pk"][i, j, v - 1]), s.reshape(1, N))
        want = (int(s[i]) * v << (64 - (j + 1) * 2)) % 2 ** 64
        assert oracle.torus_dist(ph[0], want) < 2.0 ** 30 and oracle.torus_dist(ph[1:], np.zeros(N - 1, dtype=np.uint64)).max() < 2.0 ** 30
    for i, j, v in ((0, 0, 1), (7, 2, 5), (N, 3, 7)):
        ph = oracle.trlwe_phase(np.ascontiguousarray(K["sk"][i, j, v - 1]), s.reshape(1, N))
        s_i = int(s[i]) if i < N else 2 ** 64 - 1
        dec = (s_i * v << (64 - (j + 1) * 3)) % 2 ** 64
        want = (np.uint64(0) - s) * np.uint64(dec)
        assert oracle.torus_dist(ph, want).max() < 2.0 ** 30
    for m in (0.125, -0.3):
        c = oracle.tlwe_sample(rng, oracle.double2torus(m), s, wide["sigma"])
        assert (ref.generic_keyswitch(0, c, K["pkh"], N) == oracle.trlwe_packing1_keyswitch(c, K["pk"], 2)).all()
        mine = oracle.trlwe_priv_keyswitch(c, K["sk"], 3)
        assert (ref.generic_keyswitch(1, c, K["skh"], N) == mine).all()
        want = (np.uint64(0) - s) * np.uint64(oracle.double2torus(m))
        assert oracle.torus_dist(oracle.trlwe_phase(mine, s.reshape(1, N)), want).max() < 2.0 ** 58
    # the oracle's own generator of the private key encrypts the same messages
    sk2 = oracle.gen_priv_sk_ks_key(rng, s[:6].copy(), s, 2, 2, wide["sigma"])
    ph = oracle.trlwe_phase(np.ascontiguousarray(sk2[6, 1, 2]), s.reshape(1, N))
    assert oracle.torus_dist(ph, (np.uint64(0) - s) * np.uint64(((2 ** 64 - 1) * 3 << 60) % 2 ** 64)).max() < 2.0 ** 30


def test_public_mux_and_fdfb_KS21(oracle, ref, wide):
    """public_mux (src/bootstrap.c:369-389) and full_domain_functional_bootstrap_KS21 / _KS21_2 (:391-463) vs the reference."""
    K, N, l, Bg, rng, s = _ref_keys(wide, ref), wide["N"], wide["l"], wide["Bg"], wide["rng"], wide["s"]
    p0, p1 = oracle.u64(rng.words(N)), oracle.u64(rng.words(N))
    sel = np.stack([oracle.trlwe_sample(rng, None, s.reshape(1, N), wide["sigma"]) for _ in range(l)])
    for i in range(l):
        sel[i, 1, 0] += np.uint64(1 << (64 - (i + 1) * Bg))      # selector = gadget encryption of 1 -> picks p1
    mine = oracle.public_mux(p0, p1, oracle.ks_to_dft(sel), l, Bg)
    assert oracle.torus_dist(mine, ref.public_mux(p0, p1, sel, Bg)).max() < 2.0 ** 34
    assert oracle.torus_dist(oracle.trlwe_phase(mine, s.reshape(1, N)), p1).max() < 2.0 ** 40
    lut = oracle.u64(rng.words(8))
    tv = np.repeat(lut, 2 * N // 8)
    for variant in (0, 1):
        for i in (0, 3, 5, 7):
            c = oracle.tlwe_sample(rng, (i << 61) % 2 ** 64, wide["lwe_s"], wide["lwe_sigma"])
            mine = oracle.full_domain_functional_bootstrap_KS21(tv, c, wide["bk_dft"], K["pk"], 2, l, Bg, 8, variant)
            theirs = ref.full_domain_functional_bootstrap_KS21(tv, c, K["bkh"], K["pkh"], N, 8, variant)
            # test_FDFB_KS21 (test/tests.c:1058-1092): phase within 2^58 of in[i]
            assert oracle.torus_dist(oracle.tlwe_phase(mine, s), lut[i]) < 2.0 ** 58, (variant, i)
            assert oracle.torus_dist(oracle.tlwe_phase(theirs, s), lut[i]) < 2.0 ** 58, (variant, i)
            assert oracle.torus_dist(oracle.tlwe_phase(mine, s), oracle.tlwe_phase(theirs, s)) < 2.0 ** 54, (variant, i)


def test_multivalue_phases(oracle, ref, wide):
    """multivalue_bootstrap_phase1 / phase2 (src/bootstrap.c:232-265) vs the reference; phase 2 is integer work: bit-exact
    on the same rotated accumulators."""
    K, N, l, Bg, rng, s = _ref_keys(wide, ref), wide["N"], wide["l"], wide["Bg"], wide["rng"], wide["s"]
    tb, log_tb = 4, 2
    for m in range(4):
        c = oracle.tlwe_sample(rng, oracle.double2torus(m / 8.0), wide["lwe_s"], wide["lwe_sigma"])
        mine = oracle.multivalue_bootstrap_phase1(c, wide["bk_dft"], l, Bg, tb)
        theirs = ref.multivalue_bootstrap_phase1(c, K["bkh"], N, tb)
        # the constant test vector puts coefficients exactly on digit-rounding ties, so the two FFTs pick different (equally
        # valid) decompositions and the masks diverge completely: compare by phase
        ph_m, ph_t = (np.stack([oracle.trlwe_phase(np.ascontiguousarray(x[i]), s.reshape(1, N)) for i in range(tb + 1)]) for x in (mine, theirs))
        assert oracle.torus_dist(ph_m, ph_t).max() < 2.0 ** 46
        for lut_in in ([0, 1, 2, 3], [3, 1, 0, 2], [1, 1, 1, 1]):
            out = oracle.multivalue_bootstrap_phase2(lut_in, theirs, tb, log_tb)
            assert (out == ref.multivalue_bootstrap_phase2(lut_in, theirs, tb, log_tb)).all()
    acc = oracle.u64(rng.words(N + 1))
    c2 = oracle.u64(rng.words(2 * N)).reshape(2, N)
    for scale in (1, 2, 4, 8):     # the extraction helpers (src/trlwe.c:580-622): integer, bit-exact
        assert (oracle.trlwe_mv_extract_tlwe_scaling_addto(acc, c2, scale) == ref.trlwe_mv_extract(c2, 2, scale, acc=acc)).all()
        for mode in (0, 1, 2, 3):
            a = None if mode < 2 else acc
            assert (oracle.trlwe_mv_extract(c2, mode, scale, acc=a) == ref.trlwe_mv_extract(c2, mode, scale, acc=a)).all(), (mode, scale)


def test_circuit_bootstrap_variants(oracle, ref, wide):
    """circuit_bootstrap / _2 (src/bootstrap.c:309-344, table-lookup private key switch) and _3 as a whole vs the reference."""
    K, N, l, Bg, rng, s = _ref_keys(wide, ref), wide["N"], wide["l"], wide["Bg"], wide["rng"], wide["s"]
    ks0, ks1 = oracle.gen_priv_ks_key(rng, s, s, 10, 3, wide["sigma"])
    kska3 = np.stack([ks0, ks1])
    for m in (0.25, 0.0):
        c = oracle.tlwe_sample(rng, oracle.double2torus(m), wide["lwe_s"], wide["lwe_sigma"])
        outs = {}
        for variant in (0, 1):
            mine = oracle.circuit_bootstrap(c, wide["bk_dft"], K["sk"], 3, K["pk"], 2, l, Bg, variant)
            theirs = ref.circuit_bootstrap(c, K["bkh"], l, N, K["skh"], K["pkh"], variant)
            outs[variant] = (mine, theirs)
        mine = oracle.circuit_bootstrap_3(c, wide["bk_dft"], oracle.ks_to_dft(ks0), oracle.ks_to_dft(ks1), 3, K["pk"], 2, l, Bg)
        outs[3] = (mine, ref.circuit_bootstrap(c, K["bkh"], l, N, None, K["pkh"], 3, kska_flat=kska3, bba=3))
        for variant, (mine, theirs) in outs.items():
            for q in range(2 * l):
                ph_m = oracle.trlwe_phase(np.ascontiguousarray(mine[q]), s.reshape(1, N))
                ph_t = oracle.trlwe_phase(np.ascontiguousarray(theirs[q]), s.reshape(1, N))
                h = np.uint64((1 << (64 - (q % l + 1) * Bg)) if m else 0)
                want = np.zeros(N, dtype=np.uint64)
                if q >= l:
                    want[0] = h
                else:
                    want = (np.uint64(0) - s) * h
                tol = 2.0 ** 50 if q >= l else 2.0 ** 57     # 24 / 12-bit key-switch rounding; the private switch multiplies by s
                assert oracle.torus_dist(ph_m, want).max() < tol, (variant, q)
                assert oracle.torus_dist(ph_t, want).max() < tol, (variant, q)


def test_trgsw_accumulator_bootstrap(oracle, ref, wide):
    """functional_bootstrap_trgsw_phase1 / phase2 (src/bootstrap.c:267-306)."""
    K, N, l, Bg, rng, s = _ref_keys(wide, ref), wide["N"], wide["l"], wide["Bg"], wide["rng"], wide["s"]
    lut = oracle.u64(rng.words(4))
    tv = oracle.trlwe_torus_packing(lut, 1, N)
    for m in range(4):
        c = oracle.tlwe_sample(rng, oracle.double2torus(m / 8.0), wide["lwe_s"], wide["lwe_sigma"])
        g_dft = oracle.functional_bootstrap_trgsw_phase1(c, wide["bk_dft"], l, Bg, 4)
        mine = oracle.functional_bootstrap_trgsw_phase2(g_dft, tv, l, Bg)
        acc_t, theirs = ref.functional_bootstrap_trgsw(tv, c, K["bkh"], l, 4)
        acc_m = np.stack([oracle.dft_to_torus(np.ascontiguousarray(g_dft[q, cc])) for q in range(2 * l) for cc in range(2)]).reshape(2 * l, 2, N)
        ph_m = np.stack([oracle.trlwe_phase(np.ascontiguousarray(acc_m[q]), s.reshape(1, N)) for q in range(2 * l)])
        ph_t = np.stack([oracle.trlwe_phase(np.ascontiguousarray(acc_t[q]), s.reshape(1, N)) for q in range(2 * l)])
        assert oracle.torus_dist(ph_m, ph_t).max() < 2.0 ** 46
        assert oracle.torus_dist(oracle.tlwe_phase(mine, s), lut[m]) < 2.0 ** 58
        assert oracle.torus_dist(oracle.tlwe_phase(theirs, s), lut[m]) < 2.0 ** 58
        # phase 2 multiplies the accumulator's noise (agreeing to 2^46 above) by the digits of a random 64-bit LUT
        assert oracle.torus_dist(oracle.tlwe_phase(mine, s), oracle.tlwe_phase(theirs, s)) < 2.0 ** 58


def test_tensor_product_tlwe_mul_and_fdfb_CLOT21(oracle, ref, wide):
    """trlwe_tensor_prod_FFT (src/trlwe.c:727-771), tlwe_mul (src/tlwe.c:322-332), full_domain_functional_bootstrap_CLOT21 / _2
    (src/bootstrap.c:465-517), parameters of test_FDFB_CLOT21_2 (test/tests.c:1179-1218): relinearisation key t=2, bb=20, precision 4."""
    K, N, l, Bg, rng, s = _ref_keys(wide, ref), wide["N"], wide["l"], wide["Bg"], wide["rng"], wide["s"]
    rl = oracle.gen_rl_key(rng, s, 2, 20, wide["sigma"])
    rl_dft = oracle.ks_to_dft(rl)
    precision = 4
    m1 = np.zeros(N, dtype=np.uint64); m2 = np.zeros(N, dtype=np.uint64)
    m1[0], m2[0] = 3 << 60, 2 << 60
    c1, c2 = (oracle.trlwe_sample(rng, m, s.reshape(1, N), wide["sigma"]) for m in (m1, m2))
    mine = oracle.trlwe_tensor_prod_fft(c1, c2, precision, rl_dft, 20)
    theirs = ref.trlwe_tensor_prod_FFT(c1, c2, precision, rl, 20)
    ph_m, ph_t = oracle.trlwe_phase(mine, s.reshape(1, N)), oracle.trlwe_phase(theirs, s.reshape(1, N))
    assert oracle.torus_dist(ph_m, ph_t).max() < 2.0 ** 56
    assert oracle.torus_dist(ph_m[0], (6 << 60) % 2 ** 64) < 2.0 ** 58       # 3/16 * 2/16 * 2^precision = 6/16
    t1 = oracle.tlwe_sample(rng, 3 << 60, s, wide["sigma"])
    t2 = oracle.tlwe_sample(rng, 2 << 60, s, wide["sigma"])
    mine = oracle.tlwe_mul(t1, t2, precision, K["pk"], 2, rl_dft, 20)
    theirs = ref.tlwe_mul(t1, t2, precision, K["pkh"], rl, 20)
    assert oracle.torus_dist(oracle.tlwe_phase(mine, s), oracle.tlwe_phase(theirs, s)) < 2.0 ** 57
    assert oracle.torus_dist(oracle.tlwe_phase(mine, s), (6 << 60) % 2 ** 64) < 2.0 ** 58
    lut = oracle.u64([(int(x) & 15) << 60 for x in rng.words(8)])
    tvs = np.stack([oracle.trlwe_torus_packing(lut[:4], 1, N), oracle.trlwe_torus_packing(lut[4:], 1, N)])
    for i in (0, 2, 5, 7):
        c = oracle.tlwe_sample(rng, (i << 61) % 2 ** 64, wide["lwe_s"], wide["lwe_sigma"])
        for variant, tv in ((0, tvs), (1, lut)):
            mine = oracle.full_domain_functional_bootstrap_CLOT21(tv, c, wide["bk_dft"], K["pk"], 2, rl_dft, 20, l, Bg, precision, variant)
            theirs = ref.full_domain_functional_bootstrap_CLOT21(tv, c, K["bkh"], K["pkh"], rl, 20, precision, variant)
            tol = 2.0 ** (64 - precision - 1)       # the reference test's own bound (test/tests.c:1207)
            assert oracle.torus_dist(oracle.tlwe_phase(mine, s), lut[i]) < tol, (variant, i)
            assert oracle.torus_dist(oracle.tlwe_phase(theirs, s), lut[i]) < tol, (variant, i)


@pytest.mark.parametrize("unfolding", [2, 4])
def test_functional_bootstrap_unfolded(oracle, ref, unfolding):
    """blind_rotate_unfolded through functional_bootstrap with key->unfolding > 1 (src/bootstrap.c:23-48,124-149,192-206;
    test_functional_bootstrap_unfolded, test/tests.c:1486-1530): phases agree with the reference and decrypt within 2^58."""
    rng = oracle.Rng(0xF01D + unfolding)
    N, n, l, Bg, sigma = 1024, 24, 2, 8, 2.0 ** -40
    lwe_s = oracle.gen_binary_key(rng, n)
    s = oracle.gen_binary_key(rng, N)
    su = oracle.gen_bootstrap_key_unfolded(rng, lwe_s, s, l, Bg, sigma, unfolding)
    assert su.shape == (n * (1 << unfolding) // unfolding, 2 * l, 2, N)
    h = ref.bk_unfolded_new(su, l, Bg, unfolding)
    lut = oracle.u64(rng.words(4))
    tv = oracle.trlwe_torus_packing(lut, 1, N)
    for m in range(4):
        c = oracle.tlwe_sample(rng, oracle.double2torus(m / 8.0), lwe_s, 2.0 ** -20)
        mine = oracle.functional_bootstrap_unfolded(tv, c, su, l, Bg, 4, unfolding)
        theirs = ref.functional_bootstrap(tv, c, h, 4)
        assert oracle.torus_dist(oracle.tlwe_phase(mine, s), lut[m]) < 2.0 ** 58
        assert oracle.torus_dist(oracle.tlwe_phase(theirs, s), lut[m]) < 2.0 ** 58
        assert oracle.torus_dist(oracle.tlwe_phase(mine, s), oracle.tlwe_phase(theirs, s)) < 2.0 ** 50
        if unfolding == 2:
            # the order the GPU's unfolding-2 kernel computes in (per-group TRGSW assembled in the DFT domain): the same result up to FFT rounding
            dft = oracle.functional_bootstrap_unfolded2_dft(tv, c, oracle.su_to_dft(su, l), l, Bg, 4)
            assert oracle.torus_dist(oracle.tlwe_phase(dft, s), lut[m]) < 2.0 ** 58
            assert oracle.torus_dist(oracle.tlwe_phase(dft, s), oracle.tlwe_phase(theirs, s)) < 2.0 ** 50
            assert oracle.torus_dist(oracle.tlwe_phase(dft, s), oracle.tlwe_phase(mine, s)) < 2.0 ** 34
    # multivalue_bootstrap_UBR_phase1 / phase2 (src/bootstrap.c:151-190): one phase 1, several test vectors; in the oracle the
    # pair computes exactly what the unfolded bootstrap computes
    luts = oracle.u64(rng.words(8)).reshape(2, 4)
    tvs = np.stack([oracle.trlwe_torus_packing(x, 1, N) for x in luts])
    c = oracle.tlwe_sample(rng, oracle.double2torus(3 / 8.0), lwe_s, 2.0 ** -20)
    sa = oracle.multivalue_bootstrap_UBR_phase1(c, su, l, Bg, unfolding)
    theirs = ref.multivalue_bootstrap_UBR(tvs, c, h, 4)
    for i in range(2):
        mine = oracle.multivalue_bootstrap_UBR_phase2(tvs[i], c, sa, l, Bg, unfolding, 4)
        assert (mine == oracle.functional_bootstrap_unfolded(tvs[i], c, su, l, Bg, 4, unfolding)).all()
        assert oracle.torus_dist(oracle.tlwe_phase(mine, s), luts[i][3]) < 2.0 ** 58
        assert oracle.torus_dist(oracle.tlwe_phase(mine, s), oracle.tlwe_phase(theirs[i], s)) < 2.0 ** 50
    ref.bk_free(h)
