"""Latency of small batches at N = 2048 / 4096: pbs_wide_team_kernel against pbs_kernel (run through gpurun): tools/gpu_latency_wide.py [lvl2|ufhe|set2|set3]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host, engine
name = sys.argv[1] if len(sys.argv) > 1 else "lvl2"
P = dict({"set2": ma.PARAMS_SET2, "set3": ma.PARAMS_SET3}.get(name, ma.PARAMS_LVL2))
if name == "ufhe":      # applications/multi-ciphertext-arith/src/ufhe.c:19: N = 2048, l = 6, Bg = 2^7
    P.update(l=6, Bg_bit=7)
host.seed(0x57494445)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
eng = ma.Engine(0)
key = eng.generate_bootstrap_key(rk.s[0], lk.s, P['l'], P['Bg_bit'], P['rlwe_sigma'], 1)
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
tv = host.torus_packing(lut, 1, P['N'])
sk = rk.extracted_lwe_key().s
for B in (1, 16, 64, 128, 256, 384, 512, 768, 1024):
    cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk)
    d_tv, d_ct = ma.to_device(tv[None], eng.device), ma.to_device(cts, eng.device)
    res = {}
    for label, lim in (("throughput", 0), ("wide team", 1 << 20)):
        engine.set_wide_team_max_batch(lim)
        out = eng.functional_bootstrap(key, d_tv, d_ct, 4)
        torch.cuda.synchronize()
        res[label + " out"] = ma.to_numpy(out)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                eng.functional_bootstrap(key, d_tv, d_ct, 4, out=out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 3)
        res[label] = best
    same = bool((res["throughput out"] == res["wide team out"]).all())
    ph = host.tlwe_phase(res["wide team out"], sk)
    err = np.abs((ph - lut[np.arange(B) % 4]).astype(np.int64).astype(np.float64)).max()
    print("%s l=%d B=%4d  throughput kernel %.2f ms, wide team %.2f ms (%.2fx)  identical=%s  max phase err 2^%.1f" %
          (name, P['l'], B, res["throughput"], res["wide team"], res["throughput"] / res["wide team"], same, np.log2(err + 1)))
