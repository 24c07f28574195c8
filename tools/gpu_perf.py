"""Quick GPU perf + correctness probe of the PBS kernel (run through gpurun): tools/gpu_perf.py [B] [set1|lvl2|set2|set3]"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host
from oracle import oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = dict({"set1": ma.PARAMS_SET1, "lvl2": ma.PARAMS_LVL2, "set2": ma.PARAMS_SET2, "set3": ma.PARAMS_SET3}[sys.argv[2] if len(sys.argv) > 2 else "set1"])
host.seed(0x4D4F5346)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
bk = host.gen_bootstrap_key(rk, lk, P['l'], P['Bg_bit'])
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(bk, 1, P['l'], P['Bg_bit'])
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
tv = host.torus_packing(lut, 1, P['N'])
cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk)
d_tv, d_ct = ma.to_device(tv[None], eng.device), ma.to_device(cts, eng.device)
out = eng.programmable_bootstrap(bsk, d_tv, d_ct, 3); torch.cuda.synchronize()
o = ma.to_numpy(out)
bkd = O.bk_to_dft(bk, 1, P['l'])
ok = all((o[b] == O.programmable_bootstrap(tv, cts[b], bkd, P['l'], P['Bg_bit'], 3, 0, 0)).all() for b in (0, 1, B // 2, B - 1))
ph = host.tlwe_phase(o, rk.extracted_lwe_key().s)
err = np.abs((ph - lut[np.arange(B) % 4]).astype(np.int64).astype(np.float64)).max()
ms = [eng.time_programmable_bootstrap(bsk, d_tv, d_ct, 3, 3, out=out) for _ in range(3)]
print("N=%d B=%d bit-exact-vs-oracle=%s max-phase-err=2^%.1f kernel ms=%s  -> %.1f k PBS/s" % (P['N'], B, ok, np.log2(err + 1), ["%.2f" % m for m in ms], B / min(ms)))
