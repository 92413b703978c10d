"""CPU suite: the data-parallel path with world_size 2 over gloo — sharding + SUM all-reduce of loss-pre-scaled
gradients equals one process seeing the global batch.  The per-rank "engine" here is the CPU oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from kokoro_ruslan_amd import dp
    from oracle import kokoro_oracle as O
    r, w, _ = dp.init("gloo")
    assert (r, w) == (rank, world)
    d = O.ModelDims(vocab=59, mel=20, hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=96, dec_ff=96, var_filter=32,
                    var_bins=16, max_len=300)
    P, Bf, hp = O.init_params(d, 7), O.make_buffers(d), O.StepHyper()
    glob = O.synthetic_batch(4, 32, 6, d, seed=21)                     # fixed-shape: equal valid counts per rank
    shard = dp.shard_batch(glob, rank, world)
    sync = dp.GradSync(world, bucket_elems=50_000)                     # several buckets
    G, _, _ = O.grads_of(P, Bf, shard, d, hp, loss_scale=sync.loss_scale)
    flat = torch.cat([g.reshape(-1) for g in G.values()])
    sync(flat)
    mx = dp.all_max(torch.tensor([float(rank)]))
    dp.barrier()
    if rank == 0:
        Gf, _, _ = O.grads_of(P, Bf, glob, d, hp)
        ref = torch.cat([g.reshape(-1) for g in Gf.values()])
        q.put((float((flat - ref).abs().max()), float(ref.abs().max()), float(mx)))
    dist.destroy_process_group()


def test_dp_two_ranks_equal_global_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale, mx = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err <= 2e-5 * max(scale, 1.0), (err, scale)
    assert mx == 1.0


def test_shard_batch_and_env():
    from kokoro_ruslan_amd import dp
    b = {"mel_specs": torch.arange(8.0).view(8, 1, 1), "x": torch.arange(8)}
    parts = [dp.shard_batch(b, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p["x"] for p in parts]), b["x"])
    with pytest.raises(ValueError):
        dp.shard_batch(b, 0, 3)
    assert dp.GradSync(8).loss_scale == 0.125
