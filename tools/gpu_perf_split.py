"""pbs_split_kernel under the profiler (tools/profile_r06.sh): 128 lvl2 bootstraps -- one GPU's share of configs[3] / [4] over eight -- on two CUs each, ten launches, then
a single bootstrap ten times; the one-CU kernel (pbs_wide_pair_kernel) on the same inputs for comparison.  Outputs checked by phase."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mosfhet_amd as ma
from mosfhet_amd import host, engine

P = dict(ma.PARAMS_LVL2)
host.seed(0x5317)
lk = host.LweKey(P['n'], P['lwe_sigma'])
rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
eng = ma.Engine(0)
key = eng.generate_bootstrap_key(rk.s[0], lk.s, P['l'], P['Bg_bit'], P['rlwe_sigma'], 1)
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
d_tv = ma.to_device(host.torus_packing(lut, 1, P['N'])[None], eng.device)
d_ct = ma.to_device(host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(128)], lk), eng.device)
sk = rk.extracted_lwe_key().s
for label, smax in (("two CUs per bootstrap", -1), ("one CU per bootstrap", 0)):
    engine.set_split_max_batch(smax)
    for B in (128, 1):
        out = eng.functional_bootstrap(key, d_tv, d_ct[:B], 4)
        torch.cuda.synchronize()
        err = np.abs((host.tlwe_phase(ma.to_numpy(out), sk) - lut[np.arange(B) % 4]).astype(np.int64).astype(np.float64)).max()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.functional_bootstrap(key, d_tv, d_ct[:B], 4, out=out)
        e1.record()
        torch.cuda.synchronize()
        print("%s, %3d bootstraps: %.3f ms per launch, max phase error 2^%.1f" % (label, B, e0.elapsed_time(e1) / 10, np.log2(err + 1)))
