import sys, time, numpy as np, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host
P = dict(ma.PARAMS_LVL2); B = 4096
host.seed(5)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(host.gen_bootstrap_key(rk, lk, P['l'], P['Bg_bit']), 1, P['l'], P['Bg_bit'])
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
d_tv = ma.to_device(host.torus_packing(lut, 1, P['N'])[None], eng.device)
d_ct = ma.to_device(host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk), eng.device)
out = eng.empty(B, P['N'] + 1)
for chunk in (4096, 2048, 1024, 512):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.time()
        for lo in range(0, B, chunk):
            eng.programmable_bootstrap(bsk, d_tv, d_ct[lo:lo + chunk], 3, 0, 0, out=out[lo:lo + chunk])
        torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
    print("chunk %d: %s ms" % (chunk, ["%.1f" % x for x in ts]))
