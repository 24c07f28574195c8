"""GPU perf + correctness probe of the LWE key-switch kernel: tools/gpu_perf_ks.py [B] [set1|lvl2] [device]
(`device`: the table is generated on the GPU -- no host key, no oracle comparison: for profiling runs, where the 1.2 GB lvl2 table would otherwise be
generated on the host once per counter pass)"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mosfhet_amd as ma
from mosfhet_amd import host
from oracle import oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = dict(ma.PARAMS_LVL2 if (len(sys.argv) > 2 and sys.argv[2] == "lvl2") else ma.PARAMS_SET1)
host.seed(7)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
ok_ = rk.extracted_lwe_key()
on_device = len(sys.argv) > 3 and sys.argv[3] == "device"
eng = ma.Engine(0)
if on_device:
    dk = eng.generate_keyswitch_key(lk.s, ok_.s, P['t'], P['base_bit'], P['lwe_sigma'], 7)
else:
    t0 = time.time(); ksk = host.gen_tlwe_ks_key(lk, ok_, P['t'], P['base_bit']); print("ksk gen %.1fs %.1f MB" % (time.time() - t0, ksk.nbytes / 1e6))
    dk = eng.load_keyswitch_key(ksk, P['base_bit'])
cts = host.tlwe_samples([host.double2torus((b % 8) / 8.0) for b in range(B)], ok_)
d_ct = ma.to_device(cts, eng.device)
out = eng.tlwe_keyswitch(dk, d_ct); torch.cuda.synchronize()
o = ma.to_numpy(out)
ok = None if on_device else all((o[b] == O.tlwe_keyswitch(cts[b], ksk, P['n'], P['t'], P['base_bit'])).all() for b in (0, 1, B // 2, B - 1))
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t = time.time(); eng.tlwe_keyswitch(dk, d_ct, out=out); torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
alg = B * P['N'] * P['t'] * (P['n'] + 1) * 8
print("N=%d->n=%d B=%d bit-exact=%s ms=%s -> %.1f k KS/s, algorithmic %.1f GB -> %.2f TB/s" % (P['N'], P['n'], B, ok, ["%.2f" % x for x in ts], B / min(ts), alg / 1e9, alg / min(ts) / 1e9))
