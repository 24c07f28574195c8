"""Row errors of circuit_bootstrap_3 outputs at config-4 keys on the GPU (cf. tests/golden/make_noise_lvl2_golden.py): |error at X^0| of the packing
key switch rows (the rounding term) and the rms elsewhere (key-row noise).  tools/cb_row_errors.py [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mosfhet_amd as ma
from mosfhet_amd import host
from oracle import oracle as O
O.build()
seed = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0xC4
P = dict(ma.PARAMS_LVL2); N, l, Bg, n = P["N"], P["l"], P["Bg_bit"], P["n"]
host.seed(seed)
lk = host.LweKey(n, P["lwe_sigma"]); rk = host.RlweKey(N, 1, P["rlwe_sigma"]); s = rk.s[0]
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(host.gen_bootstrap_key(rk, lk, l, Bg), 1, l, Bg)
kska = eng.load_trlwe_ks_keys(host.gen_priv_ks_key(rk, rk, 20, 2), 2)
pk = eng.generate_table_key(0, s, s, 6, 4, P["rlwe_sigma"], seed=99, compressed=True)
B = 64
cts = host.tlwe_samples([host.double2torus(0.25) for _ in range(B)], lk)
out = ma.to_numpy(eng.circuit_bootstrap_3(bsk, kska, pk, ma.to_device(cts, eng.device)))
E, rest, low = [], [], []
for b in range(B):
    for i in range(l):
        ph = O.trlwe_phase(out[b, l + i], s.reshape(1, N))
        want = np.zeros(N, dtype=np.uint64); want[0] = np.uint64(1) << np.uint64(64 - (i + 1) * Bg)
        d = (ph - want).astype(np.int64).astype(np.float64)
        E.append(abs(d[0])); rest.append(np.sqrt((d[1:] ** 2).mean()))
lowrow = []
for b in range(B):
    for i in range(l):
        ph2 = O.trlwe_phase(out[b, i], s.reshape(1, N))
        w2 = (np.uint64(0) - s) * (np.uint64(1) << np.uint64(64 - (i + 1) * Bg))
        lowrow.append(np.sqrt(((ph2 - w2).astype(np.int64).astype(np.float64) ** 2).mean()))
print("private-key-switch rows (i < l): rms 2^%.2f" % np.log2(np.sqrt((np.array(lowrow) ** 2).mean())))
# the selector product of the tests: TRGSW (.) rnd, on the GPU and through the oracle on the same TRGSW rows
rng = np.random.default_rng(1)
msg = rng.integers(0, 2 ** 64, size=N, dtype=np.uint64)
rnd = O.trlwe_sample(O.Rng(5), msg, s.reshape(1, N), P["rlwe_sigma"])
sel = eng.load_bootstrap_key_device(ma.to_device(out, eng.device), 1, l, Bg)
d_rnd = ma.to_device(rnd[None], eng.device)
dg, do = [], []
for b in range(B):
    prod = ma.to_numpy(eng.external_product(sel, b, d_rnd))[0]
    dg.append(O.torus_dist(O.trlwe_phase(prod, s.reshape(1, N)), msg))
    if b < 16:
        g_dft = O.bk_to_dft(out[b][None], 1, l)[0]
        prod_o = O.external_product(rnd, g_dft, l, Bg)
        assert (prod_o == prod).all(), b
dg = np.stack(dg)
print("selector products on the GPU (= the oracle's on the first 16): pooled rms 2^%.2f, per-output median 2^%.2f" % (np.log2(np.sqrt((dg ** 2).mean())), np.log2(np.median(np.sqrt((dg ** 2).mean(axis=1))))))
a_digits = [(((rnd[0] + np.uint64(sum(1 << (63 - i * Bg) for i in range(l)) + (1 << (63 - l * Bg)))) >> np.uint64(64 - (j + 1) * Bg)) & np.uint64((1 << Bg) - 1)).astype(np.int64) - (1 << (Bg - 1)) for j in range(l)]
print("digit rms of rnd's mask per level:", ["%.1f" % np.sqrt((d.astype(np.float64) ** 2).mean()) for d in a_digits])
E, rest = np.array(E), np.array(rest)
print("seed %#x hw %d: rms of E0 over %d rows: 2^%.2f ; rest rms 2^%.2f" % (seed, int(s.sum()), E.size, np.log2(np.sqrt((E ** 2).mean())), np.log2(np.sqrt((rest ** 2).mean()))))
# the mask words the packing key switch rounds: are their low 40 bits uniform?
acc = ma.to_numpy(eng.functional_bootstrap(bsk, ma.to_device(host.torus_packing(np.array([1 << 60, 3 << 60], dtype=np.uint64), 1, N)[None], eng.device), ma.to_device(cts, eng.device), 2))
lowbits = (acc[:, :N] + np.uint64(1 << 39)) & np.uint64((1 << 40) - 1)
dropped = lowbits.astype(np.float64) - 2.0 ** 39
print("dropped part of the mask words: rms 2^%.2f (uniform: 2^%.2f); zero low-28 bits: %.4f" % (np.log2(np.sqrt((dropped ** 2).mean())), 40 - 0.5 * np.log2(12), ((acc[:, :N] & np.uint64((1 << 28) - 1)) == 0).mean()))
