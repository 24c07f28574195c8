"""GPU perf probe of circuit_bootstrap_3 at config 4 (N=2048 l=4 Bg=2^9, packing key t=6 bb=4, private KS t=20 bb=2),
random key material (timing only): tools/gpu_perf_cb.py [B]"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N, l, Bg, n = 2048, 4, 9, 632
tb, bbb, ta, bba = 6, 4, 20, 2
rng = np.random.default_rng(1)
def rnd(*shape):
    return rng.integers(0, 2 ** 64, size=shape, dtype=np.uint64)
eng = ma.Engine(0)
t0 = time.time()
bsk = eng.load_bootstrap_key(rnd(n, 2 * l, 2, N), 1, l, Bg)
kska = eng.load_trlwe_ks_keys(rnd(2, ta, 2, N), bba)
pk = eng.load_packing1_key(rnd(N, tb, (1 << bbb) - 1, 2, N), bbb)
print("keys up %.1fs" % (time.time() - t0))
d_ct = ma.to_device(rnd(B, n + 1), eng.device)
d_lwe = ma.to_device(rnd(B, N + 1), eng.device)
d_rl = ma.to_device(rnd(B, 2, N), eng.device)
d_tv = ma.to_device(rnd(1, 2, N), eng.device)
def timeit(name, f, reps=3):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.time(); f(); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
    print("%-28s B=%d ms=%s" % (name, B, ["%.2f" % x for x in ts]))
    return min(ts)
o1 = eng.empty(B, 2, N); o2 = eng.empty(B, 2, N); o3 = eng.empty(B, 2 * l, 2, N)
timeit("wo_extract bootstrap", lambda: eng.functional_bootstrap_wo_extract(bsk, d_tv, d_ct, 2 * l, out=o1))
timeit("packing1 keyswitch", lambda: eng.trlwe_packing1_keyswitch(pk, d_lwe, out=o2))
timeit("priv_keyswitch_2", lambda: eng.trlwe_priv_keyswitch_2(kska, d_rl, out=o1))
t = timeit("circuit_bootstrap_3", lambda: eng.circuit_bootstrap_3(bsk, kska, pk, d_ct, out=o3))
print("circuit bootstraps/s: %.0f" % (B / t * 1e3))
