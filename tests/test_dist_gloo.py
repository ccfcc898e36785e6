"""world_size-2 gloo test of the multi-GPU path's plumbing on CPU: contiguous read shards + the per-read result
all-gather reproduce the single-process result. (The per-shard compute here is the CPU oracle — this process has
no GPU; on GPUs the same functions run with backend nccl = RCCL.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, n_total, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from zoe_amd import synth
    from zoe_amd.dist import all_gather_results, shard_range

    ref = synth.reference_host(500)
    first, count = shard_range(n_total, rank, world)
    reads = synth.reads_host(ref, first, count, 60)  # every rank regenerates exactly its own shard
    sc = oracle.dna_scoring(2, -5, b"N", -10, -1)
    s, st, _ = oracle.batch_score_w256(8, sc, reads, ref, fixed_len=60, threads=1)
    gs, gst = all_gather_results(torch.from_numpy(s.view(np.int32)), torch.from_numpy(st), n_total)
    # the form bench.py uses: results written into a slab, ONE asynchronous collective, views per rank
    from zoe_amd.dist import GatheredResults, ResultSlab, gather_slabs, shard_capacity

    cap = shard_capacity(n_total, world)
    slab = ResultSlab(cap)
    slab.score[:count] = torch.from_numpy(s.view(np.int32))
    slab.status[:count] = torch.from_numpy(st)
    out = GatheredResults(world, cap, [shard_range(n_total, r, world)[1] for r in range(world)])
    gather_slabs(slab, out, async_op=True).wait()
    assert torch.equal(out.scores(), gs) and torch.equal(out.statuses(), gst)
    assert torch.equal(out.score_of_rank(rank), slab.score[:count])
    if rank == 0:
        np.save(os.path.join(out_dir, "s.npy"), gs.numpy())
        np.save(os.path.join(out_dir, "st.npy"), gst.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [400, 401])
def test_two_rank_gather_equals_single_process(tmp_path, oracle, n_total):
    from zoe_amd import synth

    port = 29500 + (os.getpid() % 2000) + n_total % 7
    mp.spawn(_worker, args=(2, n_total, port, str(tmp_path)), nprocs=2, join=True)
    ref = synth.reference_host(500)
    reads = synth.reads_host(ref, 0, n_total, 60)
    sc = oracle.dna_scoring(2, -5, b"N", -10, -1)
    s, st, _ = oracle.batch_score_w256(8, sc, reads, ref, fixed_len=60, threads=2)
    assert np.array_equal(np.load(tmp_path / "s.npy").view(np.uint32), s)
    assert np.array_equal(np.load(tmp_path / "st.npy"), st)
