"""Throughput of functional_bootstrap with an unfolding-2 key against the plain key, inputs resident (run through gpurun):
tools/gpu_perf_unfold.py [B] [set1|lvl2]   (MOSFHET_HIP_UNFOLD2_DFT=0: the torus-domain assembly)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
name = sys.argv[2] if len(sys.argv) > 2 else "set1"
P = dict({"set1": ma.PARAMS_SET1, "lvl2": ma.PARAMS_LVL2}[name])
host.seed(0x554E464F)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
eng = ma.Engine(0)
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
tv = host.torus_packing(lut, 1, P['N'])
cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk)
d_tv, d_ct = ma.to_device(tv[None], eng.device), ma.to_device(cts, eng.device)
sk = rk.extracted_lwe_key().s
for label, key in (("plain", eng.generate_bootstrap_key(rk.s[0], lk.s, P['l'], P['Bg_bit'], P['rlwe_sigma'], 1)),
                   ("unfolding 2", eng.generate_bootstrap_key_unfolded(rk.s[0], lk.s, P['l'], P['Bg_bit'], P['rlwe_sigma'], 2, 2))):
    out = eng.functional_bootstrap(key, d_tv, d_ct, 4)
    torch.cuda.synchronize()
    ph = host.tlwe_phase(ma.to_numpy(out), sk)
    err = np.abs((ph - lut[np.arange(B) % 4]).astype(np.int64).astype(np.float64))
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            eng.functional_bootstrap(key, d_tv, d_ct, 4, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3)
    print("%s N=%d B=%d %-12s %.2f ms -> %.1f k/s   phase err rms 2^%.1f max 2^%.1f, beyond 2^58: %d" %
          (name, P['N'], B, label, best, B / best, np.log2(np.sqrt((err ** 2).mean()) + 1), np.log2(err.max() + 1), int((err >= 2.0 ** 58).sum())))
    key.free()
