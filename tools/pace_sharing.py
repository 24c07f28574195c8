"""What the team rendezvous of the N = 2048 bootstraps (pace_teams) costs a caller that keeps the chip SHARED: two streams, each with a queue of full-round lvl2
launches (1024 bootstraps), so that no launch ever has all its teams resident and every rendezvous that is tried runs into its bound (1 ms).

    python tools/pace_sharing.py [launches per stream]

One process per setting (the switches are read once): rendezvous off, every launch tries (MOSFHET_HIP_PACE_SKIP=0, the behaviour up to round 4), and the default
(a launch that gave up lets the next 16 skip).  Prints wall time per launch pair; one stream alone for scale."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import sys, time
import numpy as np, torch
import mosfhet_amd as ma
from mosfhet_amd import host
reps, streams = int(sys.argv[1]), int(sys.argv[2])
P = dict(ma.PARAMS_LVL2)
eng = ma.Engine(0)
host.seed(77)
lk = host.LweKey(P['n'], P['lwe_sigma']); rk = host.RlweKey(P['N'], 1, P['rlwe_sigma'])
bsk = eng.generate_bootstrap_key(rk.s[0], lk.s, P['l'], P['Bg_bit'], P['rlwe_sigma'], seed=3)
lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
d_tv = ma.to_device(host.torus_packing(lut, 1, P['N'])[None], eng.device)
d_ct = ma.to_device(host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(1024)], lk), eng.device)
ss = [torch.cuda.Stream() for _ in range(streams)]
outs = [eng.empty(1024, P['N'] + 1) for _ in ss]
def go(n):
    for _ in range(n):
        for s, o in zip(ss, outs):
            with torch.cuda.stream(s):
                eng.programmable_bootstrap(bsk, d_tv, d_ct, 3, out=o)
    torch.cuda.synchronize()
go(2)
t0 = time.perf_counter(); go(reps); dt = time.perf_counter() - t0
print("RESULT %.3f ms per round of %d launches, skip credit left %d" % (1e3 * dt / reps, streams, eng.pace_skip_credit()))
"""


def main():
    reps = sys.argv[1] if len(sys.argv) > 1 else "40"
    for name, env, streams in (("one stream, default", {}, "1"), ("two streams, rendezvous off", {"MOSFHET_HIP_PACE": "0"}, "2"),
                               ("two streams, every launch tries (MOSFHET_HIP_PACE_SKIP=0)", {"MOSFHET_HIP_PACE_SKIP": "0"}, "2"), ("two streams, default", {}, "2")):
        e = {k: v for k, v in os.environ.items() if not k.startswith("MOSFHET_HIP_PACE")}
        e.update(env)
        e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
        r = subprocess.run([sys.executable, "-c", PROBE, reps, streams], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e, cwd=ROOT)
        res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        print("%-62s %s" % (name, res[-1][7:] if res else "FAILED\n" + r.stdout[-1500:]), flush=True)


if __name__ == "__main__":
    main()
