import sys, time, numpy as np
sys.path.insert(0, '.')
import mosfhet_amd as ma
from mosfhet_amd import host, engine
from oracle import oracle as O
import torch
P = dict(ma.PARAMS_SET1)
n = 4
host.seed(1)
lk = host.LweKey(n, P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
bk = host.gen_bootstrap_key(rk, lk, P['l'], P['Bg_bit'])
eng = ma.Engine(0)
bkd = O.bk_to_dft(bk, 1, P['l'])
rng = np.random.default_rng(3)
cts = rng.integers(0, 2**64, size=(4, 2, 1024), dtype=np.uint64)
d = ma.to_device(cts, eng.device)
def check(bsk, ki, ref_ki, tag):
    out = ma.to_numpy(eng.external_product(bsk, ki, d)); out2 = ma.to_numpy(eng.external_product(bsk, ki, d))
    ok = [bool((out[i] == O.external_product(cts[i], bkd[ref_ki], P['l'], P['Bg_bit'])).all()) for i in range(4)]
    print(tag, "ki", ki, "det", bool((out == out2).all()), "ok", ok)
bsk = eng.load_bootstrap_key(bk, 1, P['l'], P['Bg_bit'])
check(bsk, 0, 0, "plain"); check(bsk, 1, 1, "plain")
# (a) permuted key: TRGSW 1 stored first
perm = np.ascontiguousarray(bk[[1, 0, 2, 3]])
bsk2 = eng.load_bootstrap_key(perm, 1, P['l'], P['Bg_bit'])
check(bsk2, 0, 1, "perm"); check(bsk2, 1, 0, "perm")
# (c) thrash caches then retry
big = torch.empty(1 << 28, dtype=torch.int64, device=eng.device); big.fill_(7); torch.cuda.synchronize()
check(bsk, 1, 1, "after-thrash"); check(bsk, 2, 2, "after-thrash")
# (d) device-resident path
d_bk = ma.to_device(bk, eng.device)
bsk3 = eng.load_bootstrap_key_device(d_bk, 1, P['l'], P['Bg_bit'])
check(bsk3, 0, 0, "dev"); check(bsk3, 1, 1, "dev"); check(bsk3, 3, 3, "dev")
