"""Pins the CPU oracle against golden vectors produced by the REAL reference (tests/golden/make_golden.py).

Integer sub-steps: bit-exact.  FFT-based ones: within the reference's own tolerances (test/tests.c) and within the
spread between the reference's own back-ends (SURVEY.md section 4 calibration: ~2^26-2^30 for one product).
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
S1 = dict(n=585, N=1024, k=1, l=2, Bg_bit=8, lwe_sigma=9.141776004202573e-5, rlwe_sigma=2.989040792967434e-8)
L2 = dict(N=2048, k=1, l=4, Bg_bit=9)


def load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("name,P", [("s1", S1), ("l2", L2)])
def test_integer_ops_bit_exact(oracle, name, P):
    g = load("integer_ops.npz")
    p, acc = g["%s_poly" % name], g["%s_acc" % name]
    l, Bg = P["l"], P["Bg_bit"]
    for i in range(l):
        assert (oracle.poly_decompose_i(p, Bg, l, i) == g["%s_decompose_i" % name][i]).all()
    assert (oracle.poly_decompose(p, Bg, l) == g["%s_decompose" % name]).all()
    for r, a in enumerate(g["%s_rot_amounts" % name]):
        assert (oracle.poly_mul_by_xai(p, int(a)) == g["%s_mul_by_xai" % name][r]).all(), a
        assert (oracle.poly_mul_by_xai_minus_1(p, int(a)) == g["%s_mul_by_xai_minus_1" % name][r]).all(), a
        assert (oracle.poly_mul_by_xai_addto(acc, p, int(a)) == g["%s_mul_by_xai_addto" % name][r]).all(), a
    for r, gen in enumerate(g["%s_gens" % name]):
        assert (oracle.poly_permute(p, int(gen)) == g["%s_permute" % name][r]).all(), gen
    c = g["%s_trlwe" % name]
    for r, idx in enumerate(g["%s_extract_idx" % name]):
        assert (oracle.trlwe_extract_tlwe(c, int(idx)) == g["%s_extract" % name][r]).all(), idx
    assert (oracle.trlwe_torus_packing(g["%s_lut" % name], 1, P["N"]) == g["%s_packing" % name]).all()


def test_scalars_bit_exact(oracle):
    g = load("integer_ops.npz")
    for ls in (10, 11, 12):
        got = np.array([oracle.torus2int(x, ls) for x in g["torus2int_x"]], dtype=np.uint64)
        assert (got == g["torus2int_%d" % ls]).all()
    got = np.array([oracle.double2torus(float(x)) for x in g["double2torus_x"]], dtype=np.uint64)
    assert (got == g["double2torus"]).all()


@pytest.mark.parametrize("name", ["toy", "s1like", "l2like"])
def test_keyswitch_bit_exact(oracle, name):
    g = load("keyswitch.npz")
    n_in, n_out, t, bb = [int(x) for x in g["%s_params" % name]]
    for c, want in zip(g["%s_in" % name], g["%s_out" % name]):
        assert (oracle.tlwe_keyswitch(np.ascontiguousarray(c), g["%s_ksk" % name], n_out, t, bb) == want).all()
    # and it decrypts: the phase moves by at most the rounding of the dropped low bits, sum_i s_i 2^(63 - t bb)
    bound = n_in * 2.0 ** (63 - t * bb) + 2.0 ** 50
    if bound < 2.0 ** 62:
        s_in, s_out = g["%s_s_in" % name], g["%s_s_out" % name]
        for c, o in zip(g["%s_in" % name][:4], g["%s_out" % name][:4]):
            assert oracle.torus_dist(oracle.tlwe_phase(o, s_out), oracle.tlwe_phase(np.ascontiguousarray(c), s_in)) < bound


@pytest.mark.parametrize("N", [1024, 2048])
def test_fft_product_within_reference_tolerance(oracle, N):
    g = load("fft_products.npz")
    a, b, exact = g["n%d_a" % N], g["n%d_b" % N], g["n%d_exact" % N]
    assert (oracle.poly_naive_mul(a, b) == exact).all()
    mine = oracle.poly_mul_fft(a, b)
    # reference test tolerance: 2^40 (test_poly_DFT_mul, test/tests.c:262)
    assert oracle.torus_dist(mine, exact).max() < 2.0 ** 40
    rt = oracle.dft_to_torus(oracle.torus_to_dft(a))
    assert oracle.torus_dist(rt, a).max() < 2.0 ** 40  # test_poly_DFT, tests.c:238
    spreads = []
    for be in ("avx512", "ffnt"):
        key = "n%d_prod_%s" % (N, be)
        if key in g:
            # the oracle must sit inside the error band of the reference's own back-ends around the exact product
            ref_err = oracle.torus_dist(g[key], exact).max()
            assert oracle.torus_dist(mine, exact).max() < 4 * ref_err + 2.0 ** 20
            spreads.append(oracle.torus_dist(mine, g[key]).max())
    assert spreads and max(spreads) < 2.0 ** 32


@pytest.mark.parametrize("name,P", [("s1", dict(S1)), ("l2", dict(L2))])
def test_external_product_within_backend_spread(oracle, name, P):
    g = load("external_product.npz")
    s, msg, trgsw, c = g["%s_key" % name], g["%s_msg" % name], g["%s_trgsw" % name], g["%s_trlwe" % name]
    mine = oracle.external_product(c, oracle.trgsw_to_dft(trgsw, 1, P["l"]), P["l"], P["Bg_bit"])
    # TRGSW(X^5) (.) TRLWE(msg): phase ~ X^5 * msg within 2^54 (test_trgsw_trlwe_mul, tests.c:424)
    want = oracle.poly_mul_by_xai(msg, 5)
    assert oracle.torus_dist(oracle.trlwe_phase(mine, s), want).max() < 2.0 ** 54
    for be in ("avx512", "ffnt"):
        key = "%s_out_%s" % (name, be)
        if key in g:
            # ciphertext-level agreement with the reference: SURVEY section 4 measured 2^26 (S1) / 2^28 (L2)
            # between the reference's own back-ends; allow 2^32
            assert oracle.torus_dist(mine, g[key]).max() < 2.0 ** 32, be
            assert oracle.torus_dist(oracle.trlwe_phase(g[key], s), want).max() < 2.0 ** 54


@pytest.mark.parametrize("case", ["short", "full"])
def test_bootstrap_phases_match_reference(oracle, case):
    g = load("bootstrap.npz")
    P = S1
    seed, n = int(g["%s_seed" % case][0]), int(g["%s_n" % case][0])
    rng = oracle.Rng(seed)
    lwe_s = oracle.gen_binary_key(rng, n)
    rlwe_s = oracle.gen_binary_key(rng, P["N"]).reshape(1, P["N"])
    bk = oracle.gen_bootstrap_key(rng, lwe_s, rlwe_s, P["l"], P["Bg_bit"], P["rlwe_sigma"])
    lut = oracle.u64(rng.words(4))
    assert (lut == g["%s_lut" % case]).all()  # the generator replays the fixture's stream
    tv = oracle.trlwe_torus_packing(lut, 1, P["N"])
    bk_dft = oracle.bk_to_dft(bk, 1, P["l"])
    out_key = rlwe_s.reshape(-1)
    cts = g["%s_cts" % case]
    tol = 2.0 ** 58 if case == "full" else 2.0 ** 42
    for be in ("avx512", "ffnt"):
        if "%s_pbs_%s" % (case, be) not in g:
            continue
        for m in range(4):
            c = np.ascontiguousarray(cts[m])
            mine = oracle.programmable_bootstrap(tv, c, bk_dft, P["l"], P["Bg_bit"], 3, 0, 0)
            ref = g["%s_pbs_%s" % (case, be)][m]
            ph_mine, ph_ref = oracle.tlwe_phase(mine, out_key), oracle.tlwe_phase(ref, out_key)
            if case == "full":
                # 585 CMUX steps: a 1-ulp FFT difference flips a gadget digit somewhere and the two masks diverge
                # (SURVEY section 4), after which the outputs are different encryptions of the same plaintext whose
                # phases differ by the bootstrap's own noise (gadget rounding, ~2^52-2^56 at SET_1).  The criterion
                # is the reference's: each phase within 2^58 of the LUT slot (tests.c:1560), and the two within 2^58.
                assert oracle.torus_dist(ph_mine, lut[m]) < 2.0 ** 58
                assert oracle.torus_dist(ph_ref, lut[m]) < 2.0 ** 58
                assert oracle.torus_dist(ph_mine, ph_ref) < 2.0 ** 58, (be, m)
            else:  # 8 CMUX steps: no divergence yet, even the ciphertexts stay close
                assert oracle.torus_dist(mine, ref).max() < 2.0 ** 36
                assert oracle.torus_dist(ph_mine, ph_ref) < 2.0 ** 42, (be, m)
            fb = oracle.functional_bootstrap(tv, c, bk_dft, P["l"], P["Bg_bit"], 4)
            assert oracle.torus_dist(oracle.tlwe_phase(fb, out_key), oracle.tlwe_phase(g["%s_fb_%s" % (case, be)][m], out_key)) < tol
        ck = np.ascontiguousarray(g["%s_ct_kappa" % case])
        mine = oracle.programmable_bootstrap(tv, ck, bk_dft, P["l"], P["Bg_bit"], 3, 3, 0)
        assert oracle.torus_dist(oracle.tlwe_phase(mine, out_key), oracle.tlwe_phase(g["%s_pbs_kappa_%s" % (case, be)], out_key)) < tol
        wo = oracle.functional_bootstrap_wo_extract(tv, np.ascontiguousarray(cts[1]), bk_dft, P["l"], P["Bg_bit"], 4)
        ph_a = oracle.trlwe_phase(wo, rlwe_s)
        ph_b = oracle.trlwe_phase(g["%s_wo_extract_%s" % (case, be)], rlwe_s)
        assert oracle.torus_dist(ph_a, ph_b).max() < tol


def test_callers_match_reference_golden(oracle):
    """The callers either side of the bootstrap against outputs of the REAL reference (both back-ends) stored in callers.npz
    (tests/golden/make_callers_golden.py; inputs and keys replayed from the seeds in tests/golden/callers_setup.py): full-domain functional
    bootstrap, multi-value bootstrap, FFT-based TRLWE key switch, automorphisms, Galois-automorphism bootstrap, the private and packing key
    switches, the unfolded bootstrap.  Integer paths bit-exact; FFT paths within the short-key ciphertext tolerances of
    tests/test_oracle_vs_reference.py, and by phase against the plaintext."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import callers_setup as S
    g = load("callers.npz")
    O = oracle
    A, GA, K, U = S.fdfb_multivalue(O), S.galois(O), S.key_switches(O), S.unfolded(O)
    assert (np.array([A["c_mv"][0], GA["cts"][3][0], K["cs"][1][0], U["cts"][3][0]], dtype=np.uint64) == g["fp"]).all()   # streams replayed exactly
    N, l, Bg = S.N, S.L, S.BG
    bk_dft = O.bk_to_dft(A["bk"], 1, l)
    fdfb = np.stack([O.full_domain_functional_bootstrap(A["tv"], c, bk_dft, A["ksk"], l, Bg, A["t"], A["bb"], 3) for c in A["cts"]])
    mv = O.multivalue_bootstrap_CLOT21(A["tv16"], A["c_mv"], bk_dft, l, Bg, 2, 8)
    out_s = A["rlwe_s"].reshape(-1)
    for m in range(8):
        assert O.torus_dist(O.tlwe_phase(fdfb[m], out_s), A["lut"][m]) < 2.0 ** 58
    ks_mine = O.trlwe_keyswitch(GA["c_ks"], O.ks_to_dft(GA["ks"]), 4, 8)
    ak_dft = O.ks_to_dft(GA["ak"])
    aut = np.stack([O.trlwe_eval_automorphism(GA["c_aut"], gen, ak_dft[(gen - 1) // 2], l, Bg) for gen in GA["gens"]])
    ga_dft = O.bk_to_dft(GA["bk"], 1, l)
    ga = np.stack([O.functional_bootstrap_ga(GA["tv"], c, ga_dft, ak_dft, l, Bg, 4) for c in GA["cts"]])
    for m in range(4):
        assert O.torus_dist(O.tlwe_phase(ga[m], GA["s"]), GA["lut"][m]) < 2.0 ** 58
    priv = O.trlwe_priv_keyswitch_2(K["ct"], O.ks_to_dft(K["ks0"]), O.ks_to_dft(K["ks1"]), 10, 3)
    pack = np.stack([O.trlwe_packing1_keyswitch(c, K["kskb"], 3) for c in K["cs"]])
    unf = np.stack([O.functional_bootstrap_unfolded(U["tv"], c, U["su"], l, Bg, 4, U["unfolding"]) for c in U["cts"]])
    if U["unfolding"] == 2:
        su_dft = O.su_to_dft(U["su"], l)
        unf_dft = np.stack([O.functional_bootstrap_unfolded2_dft(U["tv"], c, su_dft, l, Bg, 4) for c in U["cts"]])
    seen = 0
    for be in ("avx512", "ffnt"):
        if "fdfb_" + be not in g:
            continue
        seen += 1
        assert O.torus_dist(fdfb, g["fdfb_" + be]).max() < 2.0 ** 40, be
        assert O.torus_dist(mv, g["multivalue_" + be]).max() < 2.0 ** 38, be
        assert O.torus_dist(ks_mine, g["trlwe_keyswitch_" + be]).max() < 2.0 ** 34, be
        assert O.torus_dist(aut, g["automorphism_" + be]).max() < 2.0 ** 34, be
        assert O.torus_dist(ga, g["ga_" + be]).max() < 2.0 ** 42, be
        assert O.torus_dist(priv, g["priv_keyswitch_2_" + be]).max() < 2.0 ** 36, be
        assert (pack == g["packing1_" + be]).all(), be                      # table lookup: integer, bit-exact
        for m in range(4):
            ph_m, ph_r = O.tlwe_phase(unf[m], U["s"]), O.tlwe_phase(np.ascontiguousarray(g["unfolded_" + be][m]), U["s"])
            assert O.torus_dist(ph_m, U["lut"][m]) < 2.0 ** 58 and O.torus_dist(ph_r, U["lut"][m]) < 2.0 ** 58
            assert O.torus_dist(ph_m, ph_r) < 2.0 ** 50, (be, m)
            if U["unfolding"] == 2:
                # the DFT-domain assembly of the unfolding-2 kernel (oracle_ext.c:orc_blind_rotate_unfolded2_dft) against the reference's output
                ph_d = O.tlwe_phase(unf_dft[m], U["s"])
                assert O.torus_dist(ph_d, U["lut"][m]) < 2.0 ** 58 and O.torus_dist(ph_d, ph_r) < 2.0 ** 50, (be, m)
                assert O.torus_dist(ph_d, ph_m) < 2.0 ** 34, (be, m)
    assert seen >= 1


def test_reference_phase_error_distribution_pins_the_batch_criterion(oracle):
    """tests/golden/phase_error.npz = |phase - LUT slot| of 2048 SET_1 programmable bootstraps through BOTH reference builds and the oracle
    (make_phase_error_golden.py).  It pins the batch criterion the GPU tests and bench.py use (none beyond 2^60, >= 99.5 % within 2^58) to what
    the reference produces itself, and the host layer's seeded generator + the oracle reproduce the oracle column bit for bit."""
    import sys
    GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(GOLDEN, "phase_error.npz"))
    ref = {k: g["err_" + k] for k in ("avx512", "ffnt", "oracle")}
    for k, e in ref.items():
        assert e.size == 2048 and e.max() < 2.0 ** 58 and (e < 2.0 ** 57).mean() > 0.94, k          # the reference's own tail
        assert 55.8 < np.log2(np.sqrt((e ** 2).mean())) < 56.1, k                                     # sigma ~ 2^55.97: 2^58 is 4.1 sigma
    # Per SAMPLE the three disagree (a gadget digit that FFT rounding flips early in a bootstrap re-rolls the rest of its noise: a third of the
    # samples agree to 2^40, the others differ by up to 2^57) -- only the DISTRIBUTION is comparable: equal rms within 3 %, equal quartiles within 8 %
    for k in ("ffnt", "oracle"):
        assert abs(np.sqrt((ref[k] ** 2).mean()) / np.sqrt((ref["avx512"] ** 2).mean()) - 1) < 0.03, k
        for q in (0.25, 0.5, 0.75, 0.95):
            assert abs(np.quantile(ref[k], q) / np.quantile(ref["avx512"], q) - 1) < 0.08, (k, q)
    agree = np.abs(ref["avx512"] - ref["oracle"]) < 2.0 ** 40
    assert 0.2 < agree.mean() < 1.0, agree.mean()
    # reproducibility: same seed -> same keys and ciphertexts -> the oracle's outputs again (a sample of the batch; the full batch takes a minute)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_phase_error_golden as M
    from mosfhet_amd import host
    P, lk, rk, bk, lut, tv, cts = M.inputs()
    assert (cts[0] == g["first_ct"]).all() and (tv == g["tv"]).all()
    bk_dft = oracle.bk_to_dft(bk, P["k"], P["l"])
    s_out = rk.extracted_lwe_key().s
    for b in (0, 1, 777, 2047):
        out = oracle.programmable_bootstrap(tv, cts[b], bk_dft, P["l"], P["Bg_bit"], 3, 0, 0)
        assert M.dist(host.tlwe_phase(out[None], s_out), lut[[b % 4]])[0] == ref["oracle"][b], b


def test_reference_noise_at_lvl2_pins_the_n2048_criteria(oracle):
    """tests/golden/noise_lvl2.npz (make_noise_lvl2_golden.py): the reference's own phase errors at the TFHEpp lvl2 set -- 2048 programmable bootstraps
    through both of its builds, 512 functional_bootstrap_ga at n = 632, and 192 circuit_bootstrap_3 outputs (config-4 keys, the reference's own packing
    key) multiplied into a random TRLWE sample.  They pin the batch criteria of the N = 2048 GPU tests (rms within 5-15 %, maximum + 1 bit) where
    round 2 had hand-derived "97 % / 99 % within 2^58" clauses; here: the fixture is self-consistent, and the ORACLE's bootstraps at the same
    parameters and seeds have the same noise (32 of them: the oracle takes 0.1 s per lvl2 bootstrap)."""
    import sys
    from concurrent.futures import ThreadPoolExecutor
    GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(GOLDEN, "noise_lvl2.npz"))
    assert tuple(g["params"]) == (632, 2048, 4, 9, 6, 4, 20, 2)
    rms = lambda e: np.sqrt((e ** 2).mean())
    pa, pf, ga = g["pbs_avx512"], g["pbs_ffnt"], g["ga"]
    assert pa.size == pf.size == 2048 and ga.size == 512
    # lvl2 leaves 17 bits of headroom under the reference's 2^58 assertion: rms 2^39.0, max 2^40.6
    for e in (pa, pf, ga):
        assert e.max() < 2.0 ** 41.5 and 38.5 < np.log2(rms(e)) < 39.5
    assert abs(rms(pf) / rms(pa) - 1) < 0.10
    cb = 2.0 ** g["cb_log2"].astype(np.float64) - 1.0
    assert cb.shape == (24, 2048) and g["cb_rms"].shape == g["cb_max"].shape == g["cb_within58"].shape == (192,)
    assert np.allclose(np.sqrt((cb ** 2).mean(axis=1)), g["cb_rms"][:24], rtol=1e-3) and np.allclose(cb.max(axis=1), g["cb_max"][:24], rtol=1e-3)
    # ... the circuit bootstrap does not: 1 % of the reference's own product coefficients miss its 2^58 assertion (test/tests.c:992), and one output's
    # error size varies like a chi-square with few degrees of freedom (one rounding term per gadget level dominates)
    pooled = np.sqrt((g["cb_rms"] ** 2).mean())
    assert 56.3 < np.log2(pooled) < 56.7 and 2.0 ** 58.5 < g["cb_max"].max() < 2.0 ** 60 and 0.98 < g["cb_within58"].mean() < 0.995
    assert np.log2(g["cb_rms"].max() / g["cb_rms"].min()) > 2.5
    ones = g["cb_messages"] != 0
    assert abs(np.sqrt((g["cb_rms"][ones] ** 2).mean()) / np.sqrt((g["cb_rms"][~ones] ** 2).mean()) - 1) < 0.20   # selector 1 and selector 0: the same noise
    # the oracle at the same seeds
    import mosfhet_amd as ma
    from mosfhet_amd import host
    P = dict(ma.PARAMS_LVL2)
    host.seed(int(g["seed"]))
    lk = host.LweKey(P["n"], P["lwe_sigma"])
    rk = host.RlweKey(P["N"], 1, P["rlwe_sigma"])
    bk = host.gen_bootstrap_key(rk, lk, P["l"], P["Bg_bit"])
    lut = g["lut"]
    tv = host.torus_packing(lut, 1, P["N"])
    cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(2048)], lk)[:32]
    bk_dft = oracle.bk_to_dft(bk, 1, P["l"])
    with ThreadPoolExecutor(8) as pool:
        outs = list(pool.map(lambda c: oracle.programmable_bootstrap(tv, c, bk_dft, P["l"], P["Bg_bit"], 3, 0, 0), cts))
    err = np.abs((host.tlwe_phase(np.stack(outs), rk.extracted_lwe_key().s) - lut[np.arange(32) % 4]).astype(np.int64).astype(np.float64))
    assert err.max() < 2 * pa.max() and abs(rms(err) / rms(pa) - 1) < 0.35, (np.log2(err.max()), np.log2(rms(err)))
