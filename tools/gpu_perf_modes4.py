"""Two patterns under rocprofv3: 8 lvl2 bootstraps back to back, then 8 with a 1024-ciphertext extract kernel in between (tools/gpu_perf_modes3.py)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host
P = dict(ma.PARAMS_LVL2)
host.seed(3)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
eng = ma.Engine(0)
N = P['N']
bsk = eng.generate_bootstrap_key(rk.s[0], lk.s, P['l'], P['Bg_bit'], P['rlwe_sigma'], seed=5)
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
d_tv = ma.to_device(host.torus_packing(lut, 1, N)[None], eng.device)
B = 1024
d_ct = ma.to_device(host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk), eng.device)
o2 = eng.empty(B, 2, N); o1 = eng.empty(B, N + 1)
pbs = lambda: eng.functional_bootstrap_wo_extract(bsk, d_tv, d_ct, 4, out=o2)
for _ in range(8):
    pbs()
torch.cuda.synchronize()
for _ in range(8):
    pbs(); eng.trlwe_extract_tlwe(o2, 0, out=o1)
torch.cuda.synchronize()
