"""Same-box A/B of bootstrap-kernel variants.

    python tools/ab/run_ab.py build  name[:-DFLAG[,-DFLAG...]] ...     (CPU container: hipcc cross-compiles tools/ab/pbs_ab.hip per variant)
    python tools/ab/run_ab.py run    [set1|lvl2] [B] [rounds]           (GPU box: times every built variant, interleaved, checks bits vs production)

Variants are tools/ab/_build/ab_<name>.so (git-ignored, shipped by gpurun).  `base` (no flags) is always the production source as it stands.
"""
import ctypes as C
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value", "-Wno-comment"]


def build(specs, src="pbs_ab.hip", prefix="ab_"):
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        cmd = ["hipcc"] + FLAGS + [f for f in flags.split(",") if f] + [os.path.join(HERE, src), "-o", os.path.join(OUT, "%s%s.so" % (prefix, name))]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, p in procs:
        out = p.communicate()[0]
        print(name, "ok" if p.returncode == 0 else "FAILED\n" + out[-3000:])


def run(pset, B, rounds):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import mosfhet_amd as ma
    from mosfhet_amd import host, engine
    P = dict({"set1": ma.PARAMS_SET1, "lvl2": ma.PARAMS_LVL2}[pset])
    host.seed(0x4D4F5346)
    lk = host.LweKey(P["n"], P["lwe_sigma"])
    rk = host.RlweKey(P["N"], 1, P["rlwe_sigma"])
    eng = ma.Engine(0)
    bsk = eng.load_bootstrap_key(host.gen_bootstrap_key(rk, lk, P["l"], P["Bg_bit"]), 1, P["l"], P["Bg_bit"])
    lut = np.array([1 << 60, 5 << 60, 9 << 60, 13 << 60], dtype=np.uint64)
    tv = host.torus_packing(lut, 1, P["N"])
    cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk)
    d_tv, d_ct = ma.to_device(tv[None], eng.device), ma.to_device(cts, eng.device)
    want = ma.to_numpy(eng.programmable_bootstrap(bsk, d_tv, d_ct, 3))
    L = engine.lib()
    L.mosfhet_hip_bsk_device_dft.restype = C.c_void_p
    d_bk = L.mosfhet_hip_bsk_device_dft(bsk.h)
    tw = np.zeros(2 * (P["N"] // 2 - 1), dtype=np.float64)
    L.mosfhet_hip_twiddles(P["N"], tw.ctypes.data_as(C.c_void_p))
    d_tw = torch.from_numpy(tw).to(eng.device)
    libs = {}
    for path in sorted(glob.glob(os.path.join(OUT, "ab_*.so"))):
        name = os.path.basename(path)[3:-3]
        if len(sys.argv) > 5 and name not in sys.argv[5].split(","):
            continue
        libs[name] = C.CDLL(path)
    out = eng.empty(B, P["N"] + 1)
    times = {k: [] for k in libs}
    exact = {}
    for r in range(rounds):
        for name, lib in libs.items():
            ms = C.c_float()
            out.zero_()
            rc = lib.ab_pbs(C.c_void_p(d_bk), C.c_void_p(d_tw.data_ptr()), C.c_void_p(d_ct.data_ptr()), C.c_void_p(d_tv.data_ptr()), C.c_void_p(out.data_ptr()),
                            P["n"], B, 3, 3, C.byref(ms))
            assert rc == 0, (name, rc)
            times[name].append(ms.value)
            if r == 0:
                got = ma.to_numpy(out)
                exact[name] = bool((got == want).all())
                if not exact[name]:
                    ph = host.tlwe_phase(got, rk.extracted_lwe_key().s)
                    err = np.abs((ph - lut[np.arange(B) % 4]).astype(np.int64).astype(np.float64)).max()
                    exact[name] = "DIFFERS (max phase error 2^%.1f, %d of %d words differ)" % (np.log2(err + 1), int((got != want).sum()), got.size)
    base = min(times["base"]) if "base" in times else None
    for name in libs:
        t = times[name]
        print("%-28s min %.3f  med %.3f ms  %s  bit-exact-vs-production=%s" % (name, min(t), sorted(t)[len(t) // 2], ("(%+.1f %% vs base)" % (100 * (min(t) / base - 1))) if base else "", exact[name]))


def run_ep(B, rounds, grids, pset="set1", names=None):
    """external-product variants (tools/ab/_build/ep_*.so): B units against one key entry, each at the grid sizes given"""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import mosfhet_amd as ma
    from mosfhet_amd import host, engine
    if ":" in pset:   # N:l:Bg_bit
        N_, l_, bg_ = (int(x) for x in pset.split(":"))
        P = dict(ma.PARAMS_SET2, N=N_, l=l_, Bg_bit=bg_)
    else:
        P = dict({"set1": ma.PARAMS_SET1, "lvl2": ma.PARAMS_LVL2, "set2": ma.PARAMS_SET2, "set3": ma.PARAMS_SET3}[pset])
    host.seed(0x4D4F5346)
    lk = host.LweKey(4, P["lwe_sigma"])
    rk = host.RlweKey(P["N"], 1, P["rlwe_sigma"])
    eng = ma.Engine(0)
    host_bk = host.gen_bootstrap_key(rk, lk, P["l"], P["Bg_bit"])
    bsk = eng.load_bootstrap_key(host_bk, 1, P["l"], P["Bg_bit"])
    g = torch.Generator(device="cpu").manual_seed(1)
    d_in = torch.randint(-2 ** 63, 2 ** 63 - 1, (B, 2, P["N"]), dtype=torch.int64, generator=g).to(eng.device)
    d_in0 = None
    if os.environ.get("AB_CMUX"):   # the CMUX form: every variant gets the same in0 batch (its .so must be built with -DAB_CMUX=true)
        d_in0 = torch.randint(-2 ** 63, 2 ** 63 - 1, (B, 2, P["N"]), dtype=torch.int64, generator=g).to(eng.device)
    want = eng.external_product(bsk, 1, d_in) if d_in0 is None else eng.cmux(bsk, 1, d_in0, d_in)
    # the truth for a sample of units: the oracle (production is itself under test when a variant disagrees with it)
    from oracle import oracle as O
    O.build()
    bkd = O.bk_to_dft(host_bk, 1, P["l"])
    sample = sorted(set([0, 1, min(B - 1, grids[0]), min(B - 1, grids[0] + 1), B // 2, B - 1]))
    h_in = ma.to_numpy(d_in)
    truth = {b: O.external_product(h_in[b], bkd[1], P["l"], P["Bg_bit"]) for b in sample}
    w = ma.to_numpy(want)
    print("production vs oracle on units %s: %s" % (sample, all((w[b] == truth[b]).all() for b in sample)))
    L = engine.lib()
    L.mosfhet_hip_bsk_device_dft.restype = C.c_void_p
    d_row = L.mosfhet_hip_bsk_device_dft(bsk.h) + 1 * (2 * P["l"] * 2 * P["N"]) * 8
    tw = np.zeros(2 * (P["N"] // 2 - 1), dtype=np.float64)
    L.mosfhet_hip_twiddles(P["N"], tw.ctypes.data_as(C.c_void_p))
    d_tw = torch.from_numpy(tw).to(eng.device)
    libs = {os.path.basename(p)[3:-3]: C.CDLL(p) for p in sorted(glob.glob(os.path.join(OUT, "ep_*.so"))) if not names or os.path.basename(p)[3:-3] in names}
    for lib in libs.values():
        C.c_int.in_dll(lib, "ab_ep_bg_rt").value = P["Bg_bit"]
        if d_in0 is not None:
            C.c_void_p.in_dll(lib, "ab_ep_in0").value = d_in0.data_ptr()
    out = eng.empty(B, 2, P["N"])
    if os.environ.get("AB_OUT_DFT"):
        # DFT-domain results (trgsw_mul_trlwe_DFT before trlwe_from_DFT) of every variant against the FIRST one named: is a wrong unit already wrong in front of the
        # inverse transforms?
        ref = None
        dft = torch.zeros(B, 2, P["N"], dtype=torch.float64, device=eng.device)
        for name, lib in sorted(libs.items(), key=lambda kv: (kv[0] != os.environ.get("AB_OUT_DFT_REF", "sh_plain"), kv[0])):
            C.c_void_p.in_dll(lib, "ab_ep_out_dft").value = dft.data_ptr()
            for rep in range(int(os.environ["AB_OUT_DFT"])):
                dft.zero_()
                ms = C.c_float()
                rc = lib.ab_ep(C.c_void_p(d_row), C.c_void_p(d_tw.data_ptr()), C.c_void_p(d_in.data_ptr()), C.c_void_p(out.data_ptr()), B, grids[0], 1, C.byref(ms))
                assert rc == 0
                torch.cuda.synchronize()
                if ref is None:
                    ref = dft.clone()
                    print("DFT-domain reference: %s" % name)
                    break
                bad = (dft.view(torch.int64) != ref.view(torch.int64)).reshape(B, -1).any(dim=1).nonzero().flatten().cpu().numpy()
                print("   %s launch %d: %d of %d DFT-domain results differ from the reference's (first %s; blocks below 256: %d)" % (name, rep, len(bad), B, bad[:6].tolist(), int((bad % grids[0] < 256).sum())))
                if rep == 0 and len(bad):
                    # WHERE in a wrong unit: slot = register m (8) x thread t (T), per output component; the products are lane-local, so the pattern names the operand
                    T = P["N"] // 16
                    g = dft[torch.from_numpy(bad[:400]).to(dft.device)].view(torch.int64).cpu().numpy().reshape(-1, 2, 8, T, 2)
                    w = ref[torch.from_numpy(bad[:400]).to(dft.device)].view(torch.int64).cpu().numpy().reshape(-1, 2, 8, T, 2)
                    diff = (g != w).any(axis=4)                       # [units][component][m][t]
                    kinds = {}
                    for u in range(diff.shape[0]):
                        c0, c1 = diff[u, 0], diff[u, 1]
                        regs = tuple(int(x) for x in np.nonzero((c0 | c1).any(axis=1))[0])
                        waves = tuple(int(x) for x in np.nonzero([(c0 | c1)[:, :64].any(), (c0 | c1)[:, 64:].any()])[0])
                        key = ("both components" if c0.any() and c1.any() else ("component 0 only" if c0.any() else "component 1 only"),
                               "same slots in both" if (c0 == c1).all() else "different slots", "registers %s" % (regs,), "waves %s" % (waves,), "%d slots" % int((c0 | c1).sum()))
                        kinds[key] = kinds.get(key, 0) + 1
                    for key, cnt in sorted(kinds.items(), key=lambda kv: -kv[1])[:12]:
                        print("      %4d units: %s" % (cnt, ", ".join(key)))
                    u = 0
                    ts = np.nonzero((diff[u, 0] | diff[u, 1]).any(axis=0))[0]
                    print("      unit %d: wrong threads %s%s" % (bad[0], ts[:24].tolist(), " ..." if len(ts) > 24 else ""))
            C.c_void_p.in_dll(lib, "ab_ep_out_dft").value = None
        return
    unit_bytes = 2 * 2 * P["N"] * 8
    res = {}
    reps = int(os.environ.get("AB_REPS", "5"))
    for r in range(rounds):
        for name, lib in libs.items():
            for fn, gr in (("ab_ep", grids),) + ((("ab_epl", [256, 512]),) if P["N"] == 1024 else ()):
                for grid in gr:
                    ms = C.c_float()
                    out.zero_()
                    dbg = None
                    if "verify" in name:   # tools/spill_hazard verify_keys: 64 bytes per thread {used quad, true quad, quad + 1, unit counter}
                        dbg = torch.zeros(8 + grid * 1024, dtype=torch.int64, device=eng.device)
                        C.c_void_p.in_dll(lib, "ab_ep_in0").value = dbg.data_ptr()
                    if "mirror" in name:   # tools/spill_hazard: mismatch counter + samples come back through the (otherwise unused) in0 argument
                        dbg = torch.zeros(8 + grid * 512, dtype=torch.int64, device=eng.device)
                        C.c_void_p.in_dll(lib, "ab_ep_in0").value = dbg.data_ptr()
                    rc = getattr(lib, fn)(C.c_void_p(d_row), C.c_void_p(d_tw.data_ptr()), C.c_void_p(d_in.data_ptr()), C.c_void_p(out.data_ptr()), B, grid, reps, C.byref(ms))
                    assert rc == 0, (name, fn, rc)
                    ok = bool((out == want).all())
                    if dbg is not None and "verify" in name:
                        d32 = ma.to_numpy(dbg).view(np.uint32)
                        rec = d32[16:].reshape(grid * 128, 16)
                        hit = np.nonzero(rec[:, 8] != 0)[0]
                        print("   %s grid %d round %d: lanes whose key quad differed at its use from the drained re-read (over %d launches): %d; threads with a sample %d; bit-exact %s" % (name, grid, r, reps, int(d32[0]), len(hit), ok))
                        for i in hit[:10]:
                            print("      block %d thread %d quad %d unit+grid %d: used %s  true %s" % (i // 128, i % 128, rec[i, 8] - 1, rec[i, 9], " ".join("%08x" % x for x in rec[i, 0:4]), " ".join("%08x" % x for x in rec[i, 4:8])))
                        if len(hit):
                            blocks = np.unique(hit // 128)
                            print("      blocks: %d distinct, below 256: %d; quads %s; lanes per hit block (first 8) %s" % (len(blocks), int((blocks < 256).sum()), np.bincount(rec[hit, 8] - 1, minlength=6).tolist(), [int((hit // 128 == b).sum()) for b in blocks[:8]]))
                        dbg = None
                    if dbg is not None:
                        d = ma.to_numpy(dbg).view(np.uint64)
                        rec = d[8:].reshape(grid * 128, 4)
                        hit = np.nonzero((rec[:, 0] != rec[:, 1]) | (rec[:, 2] != rec[:, 3]))[0]
                        print("   %s grid %d round %d: lanes whose scratch reload differed from the LDS mirror (over %d launches): %d; threads with a sample %d; bit-exact %s" % (name, grid, r, reps, int(d[0]), len(hit), ok))
                        for i in hit[:6]:
                            print("      block %d thread %d: scratch {%016x %016x}  LDS mirror {%016x %016x}" % (i // 128, i % 128, rec[i, 0], rec[i, 2], rec[i, 1], rec[i, 3]))
                        if len(hit):
                            blocks = np.unique(hit // 128)
                            print("      blocks: %d distinct, below 256: %d, by block %% 8 %s; lanes per hit block (first 8) %s" % (len(blocks), int((blocks < 256).sum()), np.bincount(blocks % 8, minlength=8).tolist(), [int((hit // 128 == b).sum()) for b in blocks[:8]]))
                    if not ok:
                        o = ma.to_numpy(out)
                        bad = np.nonzero((o != w).any(axis=(1, 2)))[0]
                        if r == 0:
                            print("   %s grid %d: bad units by iteration (unit // grid): %s; by block %% 8: %s" % (name, grid, np.bincount(bad // grid, minlength=B // grid).tolist(), np.bincount((bad % grid) % 8, minlength=8).tolist()))
                            b0 = bad[0]
                            dw = (o[b0] != w[b0])
                            dd = (o[b0] - w[b0]).astype(np.int64)
                            print("   %s grid %d: unit %d: %d of %d words differ, per component %s, max |diff| 2^%.1f, first positions %s" % (
                                name, grid, b0, dw.sum(), dw.size, dw.sum(axis=1).tolist(), np.log2(np.abs(dd.astype(np.float64)).max() + 1), np.nonzero(dw.reshape(-1))[0][:8].tolist()))
                        if r == 0 and os.environ.get("AB_DIAGNOSE"):
                            # which inputs reproduce a wrong unit?  components a / b taken from this unit, the team's previous one or its next one
                            for b0 in bad[:int(os.environ["AB_DIAGNOSE"])]:
                                found = []
                                for da in (0, -grid, grid):
                                    for db in (0, -grid, grid):
                                        if (da or db) and 0 <= b0 + da < B and 0 <= b0 + db < B:
                                            x = np.stack([h_in[b0 + da][0], h_in[b0 + db][1]])
                                            if (O.external_product(x, bkd[1], P["l"], P["Bg_bit"]) == o[b0]).all():
                                                found.append((da // grid, db // grid))
                                zero = bool((o[b0] == 0).all())
                                print("   %s grid %d: wrong unit %d (block %d, iteration %d) equals the product of components (a, b) of the team's units at offsets %s%s" % (
                                    name, grid, b0, b0 % grid, b0 // grid, found, "; output all zero" if zero else ""))
                        ok = "False: %d units differ from production (first %s); vs oracle on the sample: %s" % (len(bad), bad[:6].tolist(), {b: bool((o[b] == truth[b]).all()) for b in sample})
                    res.setdefault((name + ":" + fn, grid), []).append((ms.value, ok))
    for (name, grid), v in res.items():
        t = [x[0] for x in v]
        print("%-20s grid %5d  min %.3f  med %.3f ms  -> %.0f GB/s (%.1f %% of 8 TB/s)  bit-exact=%s" % (name, grid, min(t), sorted(t)[len(t) // 2], B * unit_bytes / min(t) / 1e6,
                                                                                                       B * unit_bytes / min(t) / 1e6 / 80, [x[1] for x in v] if any(x[1] is not True for x in v) else True))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "build_ep":
        build(sys.argv[2:], "ep_ab.hip", "ep_")
    elif sys.argv[1] == "run_ep":
        run_ep(int(sys.argv[2]) if len(sys.argv) > 2 else 65536, int(sys.argv[3]) if len(sys.argv) > 3 else 5,
               [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "2048").split(",")], sys.argv[5] if len(sys.argv) > 5 else "set1",
               sys.argv[6].split(",") if len(sys.argv) > 6 else None)
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "set1", int(sys.argv[3]) if len(sys.argv) > 3 else 4096, int(sys.argv[4]) if len(sys.argv) > 4 else 5)
