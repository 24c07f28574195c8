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
