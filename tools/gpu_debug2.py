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
print("lwe key bits", lk.s)
bk = host.gen_bootstrap_key(rk, lk, P['l'], P['Bg_bit'])
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(bk, 1, P['l'], P['Bg_bit'])
bkd = O.bk_to_dft(bk, 1, P['l'])
print("export equal:", (bsk.export_dft() == bkd).all())
rng = np.random.default_rng(3)
cts = rng.integers(0, 2**64, size=(4, 2, 1024), dtype=np.uint64)
d = ma.to_device(cts, eng.device)
for ki in range(n):
    out = ma.to_numpy(eng.external_product(bsk, ki, d)); torch.cuda.synchronize()
    out2 = ma.to_numpy(eng.external_product(bsk, ki, d))
    ok = [bool((out[i] == O.external_product(cts[i], bkd[ki], P['l'], P['Bg_bit'])).all()) for i in range(4)]
    print("key", ki, "s_i", int(lk.s[ki]), "det", bool((out == out2).all()), "ok", ok)
print("export equal after:", (bsk.export_dft() == bkd).all())
