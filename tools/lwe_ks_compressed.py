#!/usr/bin/env python3
"""LWE key switch with a full table against the seed-compressed key of the same seed (device-generated): SET_1 (1024 -> 585, t = 5, base 2^2)
and lvl2 (2048 -> 632, t = 8, base 2^4), 4096 ciphertexts.  Usage: python tools/lwe_ks_compressed.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import mosfhet_amd as ma
    eng = ma.Engine(0)
    rng = np.random.default_rng(0)
    for name, n_in, n_out, t, bb in (("SET_1", 1024, 585, 5, 2), ("lvl2", 2048, 632, 8, 4)):
        s_in, s_out = rng.integers(0, 2, size=n_in, dtype=np.uint64), rng.integers(0, 2, size=n_out, dtype=np.uint64)
        cts = ma.to_device(rng.integers(0, 2 ** 64, size=(4096, n_in + 1), dtype=np.uint64), eng.device)
        outs = []
        for compressed in (False, True):
            key = eng.generate_keyswitch_key(s_out, s_in, t, bb, 2.0 ** -30, seed=7, compressed=compressed)
            out = eng.empty(4096, n_out + 1)
            for _ in range(3):
                eng.tlwe_keyswitch(key, cts, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                eng.tlwe_keyswitch(key, cts, out=out)
            torch.cuda.synchronize()
            ms = 1e2 * (time.perf_counter() - t0)
            outs.append(ma.to_numpy(out))
            print("%-6s %-15s key %9.1f MB   %6.3f ms per 4096  = %7.1f k key switches/s" % (name, "seed-compressed" if compressed else "full table", key.nbytes / 1e6, ms, 4096 / ms), flush=True)
            key.free()
        print("       identical results:", bool((outs[0] == outs[1]).all()))


if __name__ == "__main__":
    main()
