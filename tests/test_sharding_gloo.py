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


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE around it starts two ranks as a child torch.distributed.run.  No GPU here: each
    rank must reach the loud no-GPU exit (there is no CPU path), and the command's exit code must be the child's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: tests/test_gpu_parity.py::test_bench_launches_its_own_ranks runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600, env=env, cwd=root)
    assert r.returncode != 0
    # (the launcher ends the other rank as soon as the first has failed: one or two of the messages, and the launcher's own failure report)
    assert 1 <= r.stderr.count("bench.py needs a GPU") <= 2 and "torch.distributed.elastic" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
