import sys, time, numpy as np
sys.path.insert(0, '.')
import mosfhet_amd as ma
from mosfhet_amd import host, engine
from oracle import oracle as O
import torch
P = dict(ma.PARAMS_SET1)
n = 16
host.seed(1)
lk = host.LweKey(n, P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
bk = host.gen_bootstrap_key(rk, lk, P['l'], P['Bg_bit'])
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(bk, 1, P['l'], P['Bg_bit'])
bkd = O.bk_to_dft(bk, 1, P['l'])
rng = np.random.default_rng(3)
cts = rng.integers(0, 2**64, size=(32, 2, 1024), dtype=np.uint64)
for ki in (0, 7):
    out = ma.to_numpy(eng.external_product(bsk, ki, ma.to_device(cts, eng.device)))
    out2 = ma.to_numpy(eng.external_product(bsk, ki, ma.to_device(cts, eng.device)))
    print("key", ki, "deterministic:", (out == out2).all())
    for i in range(cts.shape[0]):
        want = O.external_product(cts[i], bkd[ki], P['l'], P['Bg_bit'])
        bad = np.argwhere(out[i] != want)
        if len(bad):
            d = O.torus_dist(out[i], want)
            print(" ct", i, "mismatches", len(bad), "first", bad[:6].tolist(), "maxdiff 2^%.1f" % np.log2(d.max()+1), "mindiff", d[d>0].min())
# timing of PBS at small batch sizes
lut = np.array([1<<61, 3<<61, 5<<61, 7<<61], dtype=np.uint64)
tv = host.torus_packing(lut, 1, P['N'])
for B in (1, 64, 1536, 4096):
    c = host.tlwe_samples([host.double2torus((b%4)/8.) for b in range(B)], lk)
    d_tv, d_ct = ma.to_device(tv[None], eng.device), ma.to_device(c, eng.device)
    o = eng.programmable_bootstrap(bsk, d_tv, d_ct, 3); torch.cuda.synchronize()
    t = time.time(); o = eng.programmable_bootstrap(bsk, d_tv, d_ct, 3); torch.cuda.synchronize(); dt = time.time()-t
    print("B", B, "n", n, "time %.3f ms -> per cmux-step per batch %.1f us" % (dt*1e3, dt*1e6/n))
