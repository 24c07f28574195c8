#!/usr/bin/env python3
"""Packing key switch (config-4 key: N = 2048, t = 6, base 2^4, 6 GB generated on the device): time per ciphertext against the batch size.
Usage: python tools/packing_batch_sweep.py [compressed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import mosfhet_amd as ma
    from mosfhet_amd import host
    eng = ma.Engine(0)
    N = 2048
    host.seed(1)
    rk = host.RlweKey(N, 1, 2.0 ** -44)
    s = rk.s[0]
    compressed = len(sys.argv) > 1 and sys.argv[1] == "compressed"
    pk = eng.generate_table_key(0, s, s, 6, 4, 2.0 ** -44, seed=5, compressed=compressed)
    print("key: %.2f GB on the device (%s)" % (pk.nbytes / 1e9, "seed-compressed" if compressed else "full rows"))
    rng = np.random.default_rng(0)
    for count in (512, 1024, 4096):
        cts = ma.to_device(rng.integers(0, 2 ** 64, size=(count, N + 1), dtype=np.uint64), eng.device)
        out = eng.empty(count, 2, N)
        for _ in range(2):
            eng.trlwe_packing1_keyswitch(pk, cts, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            eng.trlwe_packing1_keyswitch(pk, cts, out=out)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        print("count %5d: %7.3f ms  = %6.3f us per ciphertext" % (count, ms, 1e3 * ms / count), flush=True)


if __name__ == "__main__":
    main()
