import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host
from oracle import oracle as O
P = dict(ma.PARAMS_SET3); N, l, Bg = P['N'], P['l'], P['Bg_bit']
host.seed(3)
lk = host.LweKey(8, P['lwe_sigma']); rk = host.RlweKey(N, 1, P['rlwe_sigma'])
bk = host.gen_bootstrap_key(rk, lk, l, Bg)
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(bk, 1, l, Bg)
bkd = O.bk_to_dft(bk, 1, l)
r = O.Rng(11)
msg = O.u64(r.words(N))
c = O.trlwe_sample(r, msg, rk.s, P['rlwe_sigma'])
rng = np.random.default_rng(3)
for count in (1, 2, 6):
    for kind in ("sample", "random"):
        cts = np.stack([c] * count) if kind == "sample" else rng.integers(0, 2**64, size=(count, 2, N), dtype=np.uint64)
        for key_index in (0, 1):
            out = ma.to_numpy(eng.external_product(bsk, key_index, ma.to_device(cts, eng.device)))
            ok = [bool((out[i] == O.external_product(cts[i], bkd[key_index], l, Bg)).all()) for i in range(count)]
            ph = O.trlwe_phase(out[0], rk.s)
            d = O.torus_dist(ph, msg * lk.s[key_index]).max() if kind == "sample" else 0
            print(count, kind, key_index, ok, "phase err 2^%.1f" % np.log2(d + 1), "s_i", lk.s[key_index])
