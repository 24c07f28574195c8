"""What slows the lvl2 bootstrap kernel down when another kernel ran since the last bootstrap?  Per-launch durations (an event pair around every launch)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host
P = dict(ma.PARAMS_LVL2)
host.seed(3)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
eng = ma.Engine(0)
N = P['N']
bsk = eng.generate_bootstrap_key(rk.s[0], lk.s, P['l'], P['Bg_bit'], P['rlwe_sigma'], seed=5)
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
d_tv = ma.to_device(host.torus_packing(lut, 1, N)[None], eng.device)
B = 1024
d_ct = ma.to_device(host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk), eng.device)
o2 = eng.empty(B, 2, N); o1 = eng.empty(B, N + 1)
one = torch.zeros(64, dtype=torch.int64, device=eng.device)
st = torch.cuda.current_stream()
pbs = lambda: eng.functional_bootstrap_wo_extract(bsk, d_tv, d_ct, 4, out=o2)
def per_launch(name, between, reps=10):
    pbs(); torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); pbs(); e1.record(st)
        evs.append((e0, e1))
        between()
    torch.cuda.synchronize()
    d = [a.elapsed_time(b) for a, b in evs]
    print("%-52s mean %.2f  (%s)" % (name, float(np.mean(d[2:])), " ".join("%.1f" % x for x in d)), flush=True)
for rnd in range(2):
    per_launch("nothing in between", lambda: None)
    per_launch("extract kernel (1024 ciphertexts) in between", lambda: eng.trlwe_extract_tlwe(o2, 0, out=o1))
    per_launch("single-ciphertext bootstrap in between", lambda: eng.functional_bootstrap_wo_extract(bsk, d_tv, d_ct[:1], 4, out=o2[:1]))
