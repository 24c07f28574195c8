"""Soak of the production kernels: the same launch repeated many times, every output compared ON THE DEVICE with the first one (which is itself
compared with the oracle on a sample of units).  A kernel that is deterministic by construction must give the same bits every time; an intermittent
fault of the kind the round-3 / round-5 experiments chase (tools/spill_hazard, 1 - 2 % of the units of a launch on the failing builds) would show up here
at rates far below what a parity test of a few thousand units can see.

    python tools/soak.py [launches] [what,...]      what: ep_lvl2 ep_lvl2_cmux ep_set1 ep_set1_cmux ep_set1_global ep_set2 ep_set3 pbs_set1 pbs_lvl2
                                                           pbs_set1_small pbs_lvl2_small (latency kernels: 256 / 128 per launch) pbs_set2 ks_lvl2 cb3_lvl2 ga_lvl2 vec

Prints one line per case: launches, units per launch, launches whose output differed, units that differed in total.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import mosfhet_amd as ma
    from mosfhet_amd import host
    from oracle import oracle as O
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    what = sys.argv[2].split(",") if len(sys.argv) > 2 else ["ep_lvl2", "ep_lvl2_cmux", "ep_set1", "ep_set1_cmux", "ep_set1_global", "ep_set2", "ep_set3", "pbs_set1", "pbs_lvl2"]
    O.build()
    eng = ma.Engine(0)
    sets = {"set1": ma.PARAMS_SET1, "lvl2": ma.PARAMS_LVL2, "set2": ma.PARAMS_SET2, "set3": ma.PARAMS_SET3}
    keys = {}

    def key(pset, n):
        if (pset, n) not in keys:
            P = dict(sets[pset])
            host.seed(0x50AC + n)
            lk = host.LweKey(n, P["lwe_sigma"])
            rk = host.RlweKey(P["N"], 1, P["rlwe_sigma"])
            hb = host.gen_bootstrap_key(rk, lk, P["l"], P["Bg_bit"])
            keys[(pset, n)] = (P, lk, rk, hb, eng.load_bootstrap_key(hb, 1, P["l"], P["Bg_bit"]))
        return keys[(pset, n)]

    for case in what:
        if case == "vec":
            # the digit-parallel radix-integer callers (capi_vec.inc): whole gate sequences over pooled temporaries -- every repetition must give the first one's bits
            t0 = time.time()
            n, N, l, Bg, t, bb, B, d, M = 630, 2048, 6, 7, 6, 2, 4, 4, 64
            host.seed(0x0F50)
            lk = host.LweKey(n, 3.0517578125e-05)
            rk = host.RlweKey(N, 1, 5.684341886080802e-14)
            ex = rk.extracted_lwe_key()
            bsk = eng.generate_bootstrap_key(rk.s[0], lk.s, l, Bg, 5.684341886080802e-14, seed=31)
            ksk = eng.generate_keyswitch_key(lk.s, ex.s, t, bb, 3.0517578125e-05, seed=32, compressed=True)
            pksk = eng.generate_lut_packing_key(rk.s[0], ex.s, t, bb, B, 5.684341886080802e-14, seed=33)
            vec = eng.vector_ops(bsk, ksk, pksk, B)
            rngv = np.random.default_rng(3)
            def enc(v):
                msgs = [host.double2torus(float((int(x) >> (2 * i)) & 3) / (2 * B)) for i in range(d) for x in v]
                return ma.to_device(host.tlwe_samples(msgs, ex).reshape(d, len(v), N + 1), eng.device)
            a, b = enc(rngv.integers(0, 256, M)), enc(rngv.integers(0, 256, M))
            ops = {"add": lambda: vec.addsub(a, b), "sub": lambda: vec.addsub(a, b, subtract=True), "relu": lambda: vec.relu(a), "cmp": lambda: vec.cmp(a, b, True, True),
                   "sl_add": lambda: vec.sl_add(a, 1, b, 2, 16, signed=False), "mul_signed": lambda: vec.mul(a, b, 16, signed=True), "mul_unsigned": lambda: vec.mul(a, b, 16, signed=False),
                   "lut_cleartext": lambda: vec.lut_cleartext(a[:2].contiguous(), np.arange(16, dtype=np.uint64) * 7 % 256, d)}
            reps = max(2, launches // 100)
            for name, fn in ops.items():
                first = fn().clone()
                bad = sum(0 if torch.equal(fn(), first) else 1 for _ in range(reps))
                print("vec %-14s %4d calls x %3d integers: %d calls differed" % (name, reps, M, bad), flush=True)
            print("vec: %.1f s" % (time.time() - t0), flush=True)
            continue
        kind, _, rest = case.partition("_")
        pset, _, mode = rest.partition("_")
        t0 = time.time()
        if kind == "ep":
            P, lk, rk, hb, bsk = key(pset, 4)
            B = 16384 if P["N"] <= 2048 else 8192
            g = torch.Generator(device="cpu").manual_seed(7)
            d_in = torch.randint(-2 ** 63, 2 ** 63 - 1, (B, 2, P["N"]), dtype=torch.int64, generator=g).to(eng.device)
            d_in0 = torch.randint(-2 ** 63, 2 ** 63 - 1, (B, 2, P["N"]), dtype=torch.int64, generator=g).to(eng.device) if mode == "cmux" else None
            if mode == "global":   # per-unit selectors keep SET_1 off the LDS-key kernel: trgsw_mul_trlwe_DFT with one (here: the same) TRGSW_DFT per unit is too big; use a batch below 64
                B = 63
                d_in = d_in[:B].contiguous()
            run = (lambda out: eng.cmux(bsk, 1, d_in0, d_in, out=out)) if mode == "cmux" else (lambda out: eng.external_product(bsk, 1, d_in, out=out))
            first = run(None)
            torch.cuda.synchronize()
            bkd = O.bk_to_dft(hb, 1, P["l"])
            h_in, h_first = ma.to_numpy(d_in), ma.to_numpy(first)
            for b in (0, 1, B // 2, B - 1):
                if mode == "cmux":
                    h0 = ma.to_numpy(d_in0[b])
                    want = h0 + O.external_product(h_in[b] - h0, bkd[1], P["l"], P["Bg_bit"])
                else:
                    want = O.external_product(h_in[b], bkd[1], P["l"], P["Bg_bit"])
                assert (h_first[b] == want).all(), (case, b)
        elif kind in ("ks", "cb3", "ga"):
            # compositions and the table key switch at lvl2: random inputs, repeatability only (their parity is tests/test_gpu_parity.py's business)
            P, lk, rk, hb, bsk = key("lvl2", 40)
            N, l, Bg = P["N"], P["l"], P["Bg_bit"]
            s_, ex = rk.s[0], rk.extracted_lwe_key().s
            rngk = np.random.default_rng(11)
            if kind == "ks":
                ksk = eng.generate_keyswitch_key(lk.s, ex, P["t"], P["base_bit"], P["lwe_sigma"], seed=5)
                B = 4096
                d_x = ma.to_device(rngk.integers(0, 2 ** 64, size=(B, N + 1), dtype=np.uint64), eng.device)
                run = lambda out: eng.tlwe_keyswitch(ksk, d_x, out=out)
            elif kind == "cb3":
                kska = eng.load_trlwe_ks_keys(host.gen_priv_ks_key(rk, rk, 20, 2), 2)
                pk = eng.generate_table_key(0, s_, s_, 6, 4, P["rlwe_sigma"], seed=9, compressed=True)
                B = 128
                d_x = ma.to_device(rngk.integers(0, 2 ** 64, size=(B, 41), dtype=np.uint64), eng.device)
                run = lambda out: eng.circuit_bootstrap_3(bsk, kska, pk, d_x, out=out)
            else:
                bk_ga = eng.generate_bootstrap_key(s_, lk.s, l, Bg, P["rlwe_sigma"], seed=3, ga=True)
                gak = eng.generate_trlwe_ks_keys(s_, host.automorphism_key_sources(s_), l, Bg, P["rlwe_sigma"], seed=4)
                B = 1024
                d_x = ma.to_device(rngk.integers(0, 2 ** 64, size=(B, 41), dtype=np.uint64), eng.device)
                d_tv = ma.to_device(rngk.integers(0, 2 ** 64, size=(1, 2, N), dtype=np.uint64), eng.device)
                run = lambda out: eng.functional_bootstrap_ga(bk_ga, gak, d_tv, d_x, 4, out=out)
            first = run(None)
            torch.cuda.synchronize()
        else:
            P, lk, rk, hb, bsk = key(pset, sets[pset]["n"])
            B = {"small": 256 if pset == "set1" else 128}.get(mode, 4096 if pset == "set1" else 1024)
            lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
            tv = ma.to_device(host.torus_packing(lut, 1, P["N"])[None], eng.device)
            cts = ma.to_device(host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk), eng.device)
            run = lambda out: eng.programmable_bootstrap(bsk, tv, cts, 3, out=out)
            first = run(None)
            torch.cuda.synchronize()
            ph = host.tlwe_phase(ma.to_numpy(first), rk.extracted_lwe_key().s)
            assert O.torus_dist(ph, lut[np.arange(B) % 4]).max() < 2.0 ** 58, case
        n_case = launches if kind == "ep" else max(1, launches // (4 if kind == "ks" else 20))
        out = torch.empty_like(first)
        bad_launches = bad_units = 0
        for i in range(n_case):
            out.zero_()
            run(out)
            if not torch.equal(out, first):
                bad_launches += 1
                bad_units += int((out != first).reshape(B, -1).any(dim=1).sum())
        torch.cuda.synchronize()
        print("%-16s %6d launches x %5d units: %d launches differed, %d units in all (%.1f s)" % (case, n_case, B, bad_launches, bad_units, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
