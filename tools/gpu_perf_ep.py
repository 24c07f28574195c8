"""External-product kernel (trgsw_mul_trlwe_DFT + trlwe_from_DFT, src/trgsw.c:385-423) over a large batch against ONE key entry:
the HBM-bound kernel BASELINE.json's target names.  tools/gpu_perf_ep.py [B] [set1|lvl2] [cmux]"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host
from oracle import oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
P = dict({"set1": ma.PARAMS_SET1, "lvl2": ma.PARAMS_LVL2}[sys.argv[2] if len(sys.argv) > 2 else "set1"])
cmux = len(sys.argv) > 3 and sys.argv[3] == "cmux"
N, l = P['N'], P['l']
host.seed(0x4D4F5346)
lk = host.LweKey(4, P['lwe_sigma']); rk = host.RlweKey(N, 1, P['rlwe_sigma'])
bk = host.gen_bootstrap_key(rk, lk, l, P['Bg_bit'])
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(bk, 1, l, P['Bg_bit'])
rng = np.random.default_rng(1)
ct = rng.integers(0, 2**64, size=(B, 2, N), dtype=np.uint64)
d_ct = ma.to_device(ct, eng.device)
d_c0 = ma.to_device(rng.integers(0, 2**64, size=(B, 2, N), dtype=np.uint64), eng.device) if cmux else None
out = eng.empty(B, 2, N)
run = (lambda: eng.cmux(bsk, 1, d_c0, d_ct, out=out)) if cmux else (lambda: eng.external_product(bsk, 1, d_ct, out=out))
run(); torch.cuda.synchronize()
o = ma.to_numpy(out)
bkd = O.bk_to_dft(bk, 1, l)
if not cmux:
    ok = all((o[b] == O.external_product(ct[b], bkd[1], l, P['Bg_bit'])).all() for b in (0, 1, B // 2, B - 1))
else:
    ok = None
stream = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
for _ in range(int(os.environ.get("EP_GROUPS", "5"))):     # (profiles: EP_GROUPS=60 -- 300 launches, so that the kernel's own steady state outweighs the cold first ones)
    e0.record(stream)
    for _ in range(5):
        run()
    e1.record(stream); e1.synchronize()
    ms.append(e0.elapsed_time(e1) / 5)
byt = B * (2 * N * 8 * (3 if cmux else 2)) + 4 * l * N * 8
print("%s N=%d l=%d B=%d bit-exact-vs-oracle=%s kernel ms=%s -> %.2f M/s, %.0f GB/s algorithmic (%.1f %% of 8 TB/s)" % (
    "cmux" if cmux else "external_product", N, l, B, ok, ["%.3f" % m for m in ms[:8]], B / min(ms) / 1e3, byt / min(ms) / 1e6, byt / min(ms) / 1e6 / 80))
