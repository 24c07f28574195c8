"""GPU perf probe of functional_bootstrap_ga: tools/gpu_perf_ga.py [B] [set1|lvl2]"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NAME = sys.argv[2] if len(sys.argv) > 2 else 'set1'
P = dict(ma.PARAMS_LVL2 if NAME == 'lvl2' else ma.PARAMS_SET1)
host.seed(5)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
t0 = time.time(); bk = host.gen_bootstrap_key_ga(rk, lk, P['l'], P['Bg_bit']); ak = host.gen_automorphism_keyset(rk, P['l'], P['Bg_bit']); print("keygen %.1fs" % (time.time() - t0))
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(bk, 1, P['l'], P['Bg_bit']); gak = eng.load_automorphism_keys(ak, P['Bg_bit'])
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
tv = host.torus_packing(lut, 1, P['N'])
cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk)
d_tv, d_ct = ma.to_device(tv[None], eng.device), ma.to_device(cts, eng.device)
out = eng.functional_bootstrap_ga(bsk, gak, d_tv, d_ct, 4); torch.cuda.synchronize()
ph = host.tlwe_phase(ma.to_numpy(out), rk.extracted_lwe_key().s)
err = np.abs((ph - lut[np.arange(B) % 4]).astype(np.int64).astype(np.float64))
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t = time.time(); eng.functional_bootstrap_ga(bsk, gak, d_tv, d_ct, 4, out=out); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
print("GA bootstrap " + NAME + " B=%d: ms=%s -> %.1f k/s; phase err max 2^%.1f, frac<2^58 %.4f" % (B, ["%.1f" % x for x in ts], B / min(ts), np.log2(err.max() + 1), (err < 2.0**58).mean()))
