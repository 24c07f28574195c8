"""Latency of small bootstrap batches, throughput kernel vs latency (team) kernel: tools/gpu_latency.py [set1|lvl2]"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host, engine
P = dict(ma.PARAMS_LVL2 if len(sys.argv) > 1 and sys.argv[1] == 'lvl2' else ma.PARAMS_SET1)
host.seed(5)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
eng = ma.Engine(0)
bsk = eng.load_bootstrap_key(host.gen_bootstrap_key(rk, lk, P['l'], P['Bg_bit']), 1, P['l'], P['Bg_bit'])
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
d_tv = ma.to_device(host.torus_packing(lut, 1, P['N'])[None], eng.device)
for B in (1, 16, 64, 128, 256, 384, 512, 768, 1024, 2048):
    cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk)
    d_ct = ma.to_device(cts, eng.device)
    res = {}
    for name, thr in (("throughput", 0), ("latency", 1 << 30)):
        engine.set_team_max_batch(thr)
        out = eng.programmable_bootstrap(bsk, d_tv, d_ct, 3, 0, 0); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t = time.time(); eng.programmable_bootstrap(bsk, d_tv, d_ct, 3, 0, 0, out=out); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
        res[name] = (min(ts), ma.to_numpy(out))
    same = bool((res["throughput"][1] == res["latency"][1]).all())
    print("B=%5d  throughput kernel %.2f ms   latency kernel %.2f ms   identical=%s" % (B, res["throughput"][0], res["latency"][0], same))
