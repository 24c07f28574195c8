"""Same-box A/B of pbs_split_kernel settings through environment switches: each setting runs in its own process (the switches are read once), settings interleaved over
`rounds`; prints ms per launch for batches of 1, 16, 64 and 128 lvl2 bootstraps and a digest of the outputs (settings must agree bit for bit).
    python tools/split_ab.py "name:VAR=val,VAR=val" ...      (a bare name = no variables)"""
import os
import subprocess
import sys

PROBE = r"""
import sys, hashlib, numpy as np, torch
import mosfhet_amd as ma
from mosfhet_amd import host
P = dict(ma.PARAMS_LVL2)
host.seed(11)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
eng = ma.Engine(0)
key = eng.generate_bootstrap_key(rk.s[0], lk.s, P['l'], P['Bg_bit'], P['rlwe_sigma'], 1)
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
d_tv = ma.to_device(host.torus_packing(lut, 1, P['N'])[None], eng.device)
d_ct = ma.to_device(host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(128)], lk), eng.device)
h = hashlib.sha256()
res = []
for B in (1, 16, 64, 128):
    out = eng.functional_bootstrap(key, d_tv, d_ct[:B], 4)
    torch.cuda.synchronize()
    h.update(ma.to_numpy(out).tobytes())
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            eng.functional_bootstrap(key, d_tv, d_ct[:B], 4, out=out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    res.append("%d: %.3f" % (B, best))
print("RESULT", "  ".join(res), " digest", h.hexdigest()[:12])
"""

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
settings = []
for spec in sys.argv[1:]:
    name, _, env = spec.partition(":")
    settings.append((name, dict(kv.split("=") for kv in env.split(",") if kv)))
for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for name, env in settings:
        e = dict(os.environ, **env)
        e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
        r = subprocess.run([sys.executable, "-c", PROBE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e, cwd=root, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        print("%-22s %s" % (name, line[-1] if line else "FAILED: " + r.stdout[-500:]), flush=True)
