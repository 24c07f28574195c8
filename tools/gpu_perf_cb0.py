"""Timing probe of circuit_bootstrap (variant 0: one bootstrap per gadget level) and circuit_bootstrap_2 at N=2048 l=4, random key material: tools/gpu_perf_cb0.py [B]"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N, l, Bg, n = 2048, 4, 9, 632
rng = np.random.default_rng(1)
def rnd(*shape):
    return rng.integers(0, 2 ** 64, size=shape, dtype=np.uint64)
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(rnd(n, 2 * l, 2, N), 1, l, Bg)
sk = eng.load_priv_key(rnd(N + 1, 2, 3, 2, N), 2)
pk = eng.load_packing1_key(rnd(N, 2, 3, 2, N), 2)
d_ct = ma.to_device(rnd(B, n + 1), eng.device)
out = eng.empty(B, 2 * l, 2, N)
for variant in (0, 1):
    f = lambda: eng.circuit_bootstrap(bsk, sk, pk, d_ct, variant, out=out)
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.time(); f(); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
    print("circuit_bootstrap variant %d  B=%d ms=%s" % (variant, B, ["%.2f" % x for x in ts]))
d_tv2 = ma.to_device(rnd(2 * N), eng.device)
o = eng.empty(B, N + 1)
for variant in (0, 1):
    f = lambda: eng.full_domain_functional_bootstrap_KS21(bsk, pk, d_tv2, d_ct, 8, variant, out=o)
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.time(); f(); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
    print("full_domain_functional_bootstrap_KS21 variant %d  B=%d ms=%s" % (variant, B, ["%.2f" % x for x in ts]))
