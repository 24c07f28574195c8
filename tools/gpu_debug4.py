import sys, time, os, numpy as np
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
want = [[O.external_product(cts[i], bkd[ki], P['l'], P['Bg_bit']) for i in range(4)] for ki in range(n)]
d = ma.to_device(cts, eng.device)
bsk = eng.load_bootstrap_key(bk, 1, P['l'], P['Bg_bit'])
outbuf = eng.empty(4, 2, 1024)
def run(ki, reuse):
    o = eng.external_product(bsk, ki, d, out=outbuf if reuse else None)
    torch.cuda.synchronize()
    o = ma.to_numpy(o)
    ok = all((o[i] == want[ki][i]).all() for i in range(4))
    # does it equal some other key's result?
    eq = [kk for kk in range(n) if all((o[i] == want[kk][i]).all() for i in range(4))]
    return ok, eq
print("env", {k: v for k, v in os.environ.items() if 'SERIALIZE' in k or 'SDMA' in k})
for reuse in (False, True):
    seq = [0, 1, 0, 1, 2, 2, 3, 0, 1, 1]
    print("reuse_out", reuse, [(ki,) + run(ki, reuse) for ki in seq])
# PBS bit-exactness with tiny n
lut = np.array([1<<61, 3<<61, 5<<61, 7<<61], dtype=np.uint64)
tv = host.torus_packing(lut, 1, P['N'])
c = host.tlwe_samples([host.double2torus((b%4)/8.) for b in range(4)], lk)
for trial in range(3):
    o = ma.to_numpy(eng.programmable_bootstrap(bsk, ma.to_device(tv[None], eng.device), ma.to_device(c, eng.device), 3))
    print("pbs n=4 trial", trial, [bool((o[b] == O.programmable_bootstrap(tv, c[b], bkd, P['l'], P['Bg_bit'], 3, 0, 0)).all()) for b in range(4)])
