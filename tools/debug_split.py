"""debug: pbs_split_kernel against the oracle's by-component order on short keys"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mosfhet_amd as ma
from mosfhet_amd import host, engine
from oracle import oracle

eng = ma.Engine(0)
N, l, Bg = 2048, 4, 9
for n in (1, 2, 3, 8):
    r = oracle.Rng(0xCA11 + n)
    lwe_s = np.ones(n, dtype=np.uint64)
    s = oracle.gen_binary_key(r, N)
    bk = oracle.gen_bootstrap_key(r, lwe_s, s.reshape(1, N), l, Bg, 2.0 ** -45)
    bk_dft, bsk = oracle.bk_to_dft(bk, 1, l), eng.load_bootstrap_key(bk, 1, l, Bg)
    cts = np.stack([oracle.tlwe_sample(r, (i << 61) % 2 ** 64, lwe_s, 2.0 ** -25) for i in range(3)])
    tv = oracle.u64(r.words(2 * N)).reshape(2, N)
    d_tv, d_ct = ma.to_device(tv[None], eng.device), ma.to_device(cts, eng.device)
    res = {}
    for name, smax, lim in (("ref", 0, 200000), ("paired", -1, 200000), ("alone", -1, 0)):
        engine.set_split_max_batch(smax)
        engine.set_split_wait_limit(lim)
        res[name] = ma.to_numpy(eng.functional_bootstrap_wo_extract(bsk, d_tv, d_ct, 4))
        if smax:
            print("   ", name, engine.split_last_launch())
    want_ref = np.stack([oracle.functional_bootstrap_wo_extract(tv, c, bk_dft, l, Bg, 4) for c in cts])
    with oracle.product_order("by_component"):
        want_bc = np.stack([oracle.functional_bootstrap_wo_extract(tv, c, bk_dft, l, Bg, 4) for c in cts])
    print("n=%d: ref==oracle_ref %s; paired==bc %s; alone==bc %s; paired==alone %s; paired==oracle_ref %s" % (
        n, (res["ref"] == want_ref).all(), (res["paired"] == want_bc).all(), (res["alone"] == want_bc).all(), (res["paired"] == res["alone"]).all(), (res["paired"] == want_ref).all()))
    for name in ("paired", "alone"):
        d = (res[name] != want_bc)
        if d.any():
            dd = np.abs((res[name] - want_bc).astype(np.int64).astype(np.float64))
            print("    %s: differing words per (ct, comp): %s, max |diff| 2^%.1f; vs ref-order: %s" % (name, d.sum(axis=2).tolist(), np.log2(dd.max() + 1), (res[name] != want_ref).sum(axis=2).tolist()))
