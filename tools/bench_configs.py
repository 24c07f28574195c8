#!/usr/bin/env python3
"""Supplementary throughput lines for BASELINE.json configs[2..4] (bench.py itself measures configs[1], the headline metric).
Same timing discipline as bench.py (barrier + synchronize around K steps, max over ranks, mosfhet_amd/shard.py); one JSON line per
workload on rank 0.  Usage: python tools/bench_configs.py [--steps K] [--warmup W] [--only name]   (torch.distributed.run for N > 1)

  lvl2_pbs        configs[2]: 4096 programmable bootstraps per GPU at N=2048 l=4 n=632
  circuit         configs[3]: circuit_bootstrap_3, batch of 1024 split over the GPUs (strong scaling), packing key t=6 bb=4 (6 GB,
                  generated on the device), private key t=20 bb=2
  fdfb            configs[4]: full_domain_functional_bootstrap (precision 3) at N=2048, 1024 per GPU
  multivalue      configs[4]: multivalue_bootstrap_CLOT21, 8 LUTs of 2 slots, at N=2048, 1024 per GPU
  ga              configs[4]: functional_bootstrap_ga at N=2048 (automorphism keys 256 MiB), 1024 per GPU
  keyswitch_lvl2  LWE key switch N=2048 -> n=632, t=8 bb=4 (1.2 GB table), 4096 per GPU

  --in-api-devices 0,1,...   the same configs[3] / [4] workloads through the drop-in C API (host structs, *_batch entry points) with several devices behind
                  it in ONE process (mosfhet_set_devices: the library shards every batch and replicates the keys device to device): tools/in_api_devices.c
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--only", default=None)
    ap.add_argument("--share", type=int, default=1, metavar="S",
                    help="size every batch as ONE GPU's share when the configs' batches are split over S GPUs (strong scaling: 1024 / S circuit, FDFB, multi-value and "
                         "Galois bootstraps, 4096 / S lvl2 bootstraps) -- what a GPU of an S-GPU node would run, measured on this one")
    ap.add_argument("--in-api-devices", default=None, metavar="IDS",
                    help="instead: configs[3] / [4] through the drop-in C API in ONE process with these devices behind it (mosfhet_set_devices; e.g. 0,1,2,3,4,5,6,7, "
                         "or 0,0 for two contexts on one GPU): compiles and runs tools/in_api_devices.c and relays its JSON lines")
    args = ap.parse_args()
    if args.in_api_devices:
        import subprocess
        from mosfhet_amd import build
        build.build()
        exe = os.path.join(ROOT, "tools", "in_api_devices_bin")
        libdir = os.path.join(ROOT, "mosfhet_amd")
        subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "in_api_devices.c"), "-o", exe, "-L" + libdir, "-lmosfhet_hip", "-lm",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
        sys.exit(subprocess.call([exe, args.in_api_devices, str(args.steps)]))
    import torch
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if not torch.cuda.is_available():
        sys.exit("needs a GPU: mosfhet_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import mosfhet_amd as ma
    from mosfhet_amd import host, shard
    P = dict(ma.PARAMS_LVL2)
    N, l, Bg, n = P["N"], P["l"], P["Bg_bit"], P["n"]
    host.seed(0x4D4F5346)
    lk = host.LweKey(n, P["lwe_sigma"])
    rk = host.RlweKey(N, 1, P["rlwe_sigma"])
    eng = ma.Engine(local_rank)
    bsk = eng.load_bootstrap_key(host.gen_bootstrap_key(rk, lk, l, Bg), 1, l, Bg)
    s = rk.s[0]
    out_s = rk.extracted_lwe_key().s
    host.seed(0x4D4F5346 + 1 + rank)

    def emit(name, metric, unit_count, step, scaling, workload, check=None):
        if args.only and args.only != name:
            return
        step()
        torch.cuda.synchronize()
        ok = check() if check else None
        for _ in range(args.warmup):
            step()
        elapsed = shard.timed_region(step, args.steps, sync=torch.cuda.synchronize, device=eng.device)
        if rank == 0:
            total = unit_count * args.steps * (world if scaling == "weak" else 1)
            print(json.dumps({"metric": metric, "value": total / elapsed, "unit": "units/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
                              "data": "synthetic", "config": {"workload": workload + ("" if args.share == 1 else " -- sized as ONE GPU's share of %d: %d units per step" % (args.share, unit_count)),
                                                              "name": name}, "decrypts": ok}), flush=True)

    lut4 = np.array([host.double2torus(x) for x in (0.0625, 0.3125, -0.1875, 0.4375)], dtype=np.uint64)

    # ---- configs[2] ----
    B = 4096 // args.share
    cts = host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B)], lk)
    d_ct = ma.to_device(cts, eng.device)
    d_tv = ma.to_device(host.torus_packing(lut4, 1, N)[None], eng.device)
    d_out = eng.empty(B, N + 1)

    def chk_pbs():
        ph = host.tlwe_phase(ma.to_numpy(d_out), out_s)
        d = np.abs((ph - lut4[np.arange(B) % 4]).astype(np.int64).astype(np.float64))
        return bool(d.max() < 2.0 ** 58)
    emit("lvl2_pbs", "programmable bootstraps/sec, N=2048 k=1 l=4", B, lambda: eng.programmable_bootstrap(bsk, d_tv, d_ct, 3, 0, 0, out=d_out), "weak",
         "batch of 4096 programmable bootstraps per GPU, TFHEpp lvl2 n=632 N=2048 l=4 Bg=2^9 (BASELINE.json configs[2])", chk_pbs)

    # ---- configs[3] ----
    if not args.only or args.only == "circuit":
        kska = eng.load_trlwe_ks_keys(host.gen_priv_ks_key(rk, rk, 20, 2), 2)
        pk = eng.generate_table_key(0, s, s, 6, 4, P["rlwe_sigma"], seed=99)   # 6 GB of rows in HBM (seed-compressed: 3 GB, masks regenerated in the kernel, same results, switches 25 % slower)
        lo, hi = shard.shard_bounds(1024, rank, world) if args.share == 1 else shard.shard_bounds(1024, 0, args.share)
        d_cb_in = ma.to_device(host.tlwe_samples([host.double2torus(0.25 * (b & 1)) for b in range(lo, hi)], lk), eng.device)
        d_cb_out = eng.empty(hi - lo, 2 * l, 2, N)
        emit("circuit", "circuit bootstraps/sec (circuit_bootstrap_3), N=2048 l=4", 1024 if args.share == 1 else hi - lo, lambda: eng.circuit_bootstrap_3(bsk, kska, pk, d_cb_in, out=d_cb_out),
             "strong", "circuit_bootstrap_3, batch of 1024 split over the GPUs, packing key t=6 bb=4 (6 GB), private key t=20 bb=2 (BASELINE.json configs[3])")
        pk.free()
        kska.free()

    # ---- configs[4] ----
    B5 = 1024 // args.share
    if not args.only or args.only in ("fdfb", "multivalue", "keyswitch_lvl2"):
        ksk = eng.load_keyswitch_key(host.gen_tlwe_ks_key(lk, rk.extracted_lwe_key(), P["t"], P["base_bit"]), P["base_bit"])
        lut8 = np.array([host.double2torus(((3 * i + 1) % 8) / 8.0) for i in range(8)], dtype=np.uint64)
        d_tv8 = ma.to_device(host.torus_packing_many_lut(lut8, 1, N, 4, 2)[None], eng.device)
        d_in5 = ma.to_device(host.tlwe_samples([(b % 8) << 61 for b in range(B5)], lk), eng.device)
        d_o5 = eng.empty(B5, N + 1)

        def chk_fdfb():
            ph = host.tlwe_phase(ma.to_numpy(d_o5), out_s)
            return bool(np.abs((ph - lut8[np.arange(B5) % 8]).astype(np.int64).astype(np.float64)).max() < 2.0 ** 58)
        emit("fdfb", "full-domain functional bootstraps/sec, N=2048", B5, lambda: eng.full_domain_functional_bootstrap(bsk, ksk, d_tv8, d_in5, 3, out=d_o5), "weak",
             "full_domain_functional_bootstrap, precision 3, 1024 per GPU, N=2048 (BASELINE.json configs[4])", chk_fdfb)
        lut16 = np.array([host.double2torus(((5 * i + 3) % 16) / 16.0) for i in range(16)], dtype=np.uint64)
        d_tvm = ma.to_device(host.torus_packing_many_lut(lut16, 1, N, 2, 8)[None], eng.device)
        d_inm = ma.to_device(host.tlwe_samples([host.double2torus((b % 2) / 4.0) for b in range(B5)], lk), eng.device)
        d_om = eng.empty(B5, 8, N + 1)
        emit("multivalue", "multi-value bootstraps/sec (8 LUTs per blind rotation), N=2048", B5,
             lambda: eng.multivalue_bootstrap_CLOT21(bsk, d_tvm, d_inm, 2, 8, out=d_om), "weak",
             "multivalue_bootstrap_CLOT21, torus_base 2, 8 LUTs, 1024 per GPU, N=2048 (BASELINE.json configs[4])")
        d_ks_in = ma.to_device(np.random.default_rng(1).integers(0, 2 ** 64, size=(4096, N + 1), dtype=np.uint64), eng.device)
        d_ks_out = eng.empty(4096, n + 1)
        emit("keyswitch_lvl2", "LWE key switches/sec, N=2048 -> n=632, t=8 bb=4", 4096, lambda: eng.tlwe_keyswitch(ksk, d_ks_in, out=d_ks_out), "weak",
             "tlwe_keyswitch, 4096 per GPU, 1.2 GB table")
        ksk.free()
    if not args.only or args.only == "ga":
        n_ga = n
        bk_ga = eng.load_bootstrap_key(host.gen_bootstrap_key_ga(rk, lk, l, Bg), 1, l, Bg)
        gak = eng.load_automorphism_keys(host.gen_automorphism_keyset(rk, l, Bg), Bg)
        d_inga = ma.to_device(host.tlwe_samples([host.double2torus((b % 4) / 8.0) for b in range(B5)], lk), eng.device)
        d_oga = eng.empty(B5, N + 1)
        emit("ga", "Galois-automorphism functional bootstraps/sec, N=2048", B5, lambda: eng.functional_bootstrap_ga(bk_ga, gak, d_tv, d_inga, 4, out=d_oga), "weak",
             "functional_bootstrap_ga, 1024 per GPU, N=2048 n=%d (BASELINE.json configs[4])" % n_ga)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
