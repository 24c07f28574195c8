"""N > 1 path on CPU: two gloo processes shard a batch exactly as bench.py shards it across GPUs -- contiguous
slices, replicated keys, no data-path collective -- and the gathered result equals the single-process result.
The per-unit work here is the oracle's key switch (the HIP kernels need a GPU); what is under test is the
partitioning, the replication by seed, the timed-region reduction and the result gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mosfhet_amd import shard


def test_shard_bounds_cover_and_are_contiguous():
    for count in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            pieces = [shard.shard_bounds(count, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == count
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in pieces]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, count, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    # replicated keys: every rank derives the SAME key from the same seed (bench.py does this per GPU)
    rng = O.Rng(99)
    n_in, n_out, t, bb = 48, 12, 3, 2
    s_in, s_out = O.gen_binary_key(rng, n_in), O.gen_binary_key(rng, n_out)
    ksk = O.gen_tlwe_ks_key(rng, s_in, s_out, t, bb, 2.0 ** -40)
    cts = np.stack([O.tlwe_sample(rng, O.double2torus((b % 8) / 8.0), s_in, 2.0 ** -40) for b in range(count)])
    lo, hi = shard.shard_bounds(count, rank, world)
    mine = cts[lo:hi]
    out = np.zeros((hi - lo, n_out + 1), dtype=np.uint64)

    def step():
        for i in range(hi - lo):
            out[i] = O.tlwe_keyswitch(np.ascontiguousarray(mine[i]), ksk, n_out, t, bb)

    elapsed = shard.timed_region(step, 2)
    assert elapsed > 0
    full = shard.gather_rows(torch.from_numpy(out.view(np.int64)), count)
    if rank == 0:
        want = np.stack([O.tlwe_keyswitch(np.ascontiguousarray(c), ksk, n_out, t, bb) for c in cts])
        ret["ok"] = bool((full.numpy().view(np.uint64) == want).all())
        ret["elapsed"] = elapsed
    dist.destroy_process_group()


@pytest.mark.parametrize("count", [9, 16])
def test_two_rank_sharded_batch_matches_single_process(count):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, count, ret), nprocs=2, join=True)
    assert ret["ok"]
