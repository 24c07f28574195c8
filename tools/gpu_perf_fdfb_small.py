"""Where do 128 full-domain functional bootstraps at lvl2 (one GPU's share of configs[4] over eight) spend their time?  The composition, its two blind rotations
alone, its key switch alone (events on the launch stream)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mosfhet_amd as ma
from mosfhet_amd import host

P = dict(ma.PARAMS_LVL2)
N, l, Bg, B = P["N"], P["l"], P["Bg_bit"], int(sys.argv[1]) if len(sys.argv) > 1 else 128
host.seed(0xFDFB)
lk = host.LweKey(P["n"], P["lwe_sigma"])
rk = host.RlweKey(N, 1, P["rlwe_sigma"])
out_s = rk.extracted_lwe_key().s
eng = ma.Engine(0)
bsk = eng.generate_bootstrap_key(rk.s[0], lk.s, l, Bg, P["rlwe_sigma"], 1)
ksk = eng.generate_keyswitch_key(lk.s, out_s, P["t"], P["base_bit"], P["lwe_sigma"], seed=7, compressed=True)
lut8 = np.array([host.double2torus(((3 * i + 1) % 8) / 8.0) for i in range(8)], dtype=np.uint64)
d_tv8 = ma.to_device(host.torus_packing_many_lut(lut8, 1, N, 4, 2)[None], eng.device)
d_in = ma.to_device(host.tlwe_samples([(b % 8) << 61 for b in range(B)], lk), eng.device)
d_o = eng.empty(B, N + 1)
d_big = ma.to_device(np.random.default_rng(1).integers(0, 2 ** 64, size=(B, N + 1), dtype=np.uint64), eng.device)
d_ks = eng.empty(B, P["n"] + 1)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


print("%d inputs: full_domain_functional_bootstrap %.3f ms; one functional_bootstrap %.3f ms; tlwe_keyswitch N -> n %.3f ms" % (
    B, timed(lambda: eng.full_domain_functional_bootstrap(bsk, ksk, d_tv8, d_in, 3, out=d_o)), timed(lambda: eng.functional_bootstrap(bsk, d_tv8, d_in, 4, out=d_o)),
    timed(lambda: eng.tlwe_keyswitch(ksk, d_big, out=d_ks))))
