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


def _ragged(batch, seed):
    """Per-sample valid lengths (as real, padded batches have): different valid-element counts per shard."""
    g = torch.Generator().manual_seed(seed)
    b = {k: v.clone() for k, v in batch.items()}
    B, T = b["mel_specs"].shape[:2]
    P = b["phoneme_indices"].shape[1]
    b["mel_lengths"] = torch.randint(T // 3, T + 1, (B,), generator=g)
    b["phoneme_lengths"] = torch.randint(2, P + 1, (B,), generator=g)
    b["mel_lengths"][0], b["phoneme_lengths"][0] = T, P
    return b


def _worker_ragged(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from kokoro_ruslan_amd import dp, spec
    from oracle import kokoro_oracle as O
    dp.init("gloo")
    dkw = dict(vocab=59, mel=20, hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=96, dec_ff=96, var_filter=32, var_bins=16, max_len=300)
    d = O.ModelDims(**dkw)
    # the engine-side objects of a ragged data-parallel step: ONE exchange carries both collectives (the loss normalisers
    # between the loss forward and the loss backward, the gradient buckets in backward order over the arena layout)
    ex = dp.BucketedExchange(spec.ModelDims(**dkw), world, backend="dist")
    names, shapes, offset, total = spec.arena_layout(spec.ModelDims(**dkw))
    P, Bf, hp = O.init_params(d, 7), O.make_buffers(d), O.StepHyper()
    glob = _ragged(O.synthetic_batch(4, 32, 6, d, seed=21), 5)
    shard = dp.shard_batch(glob, rank, world)
    Pg = {n: p.detach().clone().requires_grad_(True) for n, p in P.items()}
    sums, counts = O.loss_sums(O.forward(Pg, Bf, shard, d, None), shard, hp)
    acc = torch.stack([s.detach().double() for s in sums] + [c.double() for c in counts])        # the engine's loss_acc
    local_counts = acc[5:].clone()
    max_dur = shard["phoneme_durations"].max().reshape(1)
    assert not ex.capturable                                            # (torch.distributed backend: eager; RCCL: captured)
    ex.loss_sync(acc, max_dur)                                          # global sums / counts, global max duration
    total_loss = O.losses_from_sums(sums, list(acc[5:]), hp)[0]         # local sums over GLOBAL counts, no 1/world
    total_loss.backward()
    arena = torch.zeros(total)                                          # the gradient arena as the engine lays it out
    for n, p in Pg.items():
        if p.grad is not None:
            arena[offset[n]:offset[n] + p.numel()] = p.grad.reshape(-1)
    ex.begin_step()
    for tag in ex.plan:                                                  # bucket by bucket, in the order the backward releases them
        ex.reduce(arena, tag)
    assert ex.issued == list(ex.plan)
    flat = torch.cat([arena[offset[n]:offset[n] + p.numel()] for n, p in Pg.items()])
    if rank == 0:
        Gf, lf, _ = O.grads_of(P, Bf, glob, d, hp)
        ref = torch.cat([g.reshape(-1) for g in Gf.values()])
        glob_losses = O.losses_from_sums([a for a in acc[:5].float()], list(acc[5:]), hp)
        q.put((float((flat - ref).abs().max()), float(ref.abs().max()), [float(x) for x in glob_losses], [float(x) for x in lf],
               bool((local_counts != acc[5:] / world).any()), int(max_dur), int(glob["phoneme_durations"].max())))
    dist.destroy_process_group()


def test_dp_ragged_shards_use_global_loss_normalisers():
    """Ragged shards: SUM of (local sums / GLOBAL counts) gradients == global-batch gradients, and the reduced sums give
    the global-batch losses (what kk_losses_finalize computes from the reduced accumulator on the GPU)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale, gl, lf, ragged, md, md_ref = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ragged, "the shards must really have different valid counts"
    assert err <= 2e-5 * max(scale, 1.0), (err, scale)
    np.testing.assert_allclose(gl, lf, rtol=1e-5, atol=1e-6)
    assert md == md_ref


def test_sampler_steps_are_equal_across_ranks():
    from kokoro.data.cached import FixedBatchSampler, step_groups
    for world in (2, 3, 8):
        per_rank = [FixedBatchSampler(103, 4, True, r, world, seed=1).batches() for r in range(world)]
        assert len({len(b) for b in per_rank}) == 1, "every rank must run the same number of steps"
        groups = step_groups(FixedBatchSampler(103, 4, True, 0, world, seed=1).global_batches(), world)
        assert len(groups) == len(per_rank[0])
        for s, g in enumerate(groups):
            assert [per_rank[r][s] for r in range(world)] == g


def test_shard_batch_and_env():
    from kokoro_ruslan_amd import dp
    b = {"mel_specs": torch.arange(8.0).view(8, 1, 1), "x": torch.arange(8)}
    parts = [dp.shard_batch(b, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p["x"] for p in parts]), b["x"])
    with pytest.raises(ValueError):
        dp.shard_batch(b, 0, 3)
    assert dp.GradSync(8).loss_scale == 0.125


# ---- bucketed in-step exchange: the engine-side plan and range logic (dp.bucket_plan / dp.BucketedExchange) -----------
def test_bucket_plan_partitions_the_arena_in_backward_order():
    from kokoro_ruslan_amd import dp, spec
    for dims in (spec.ModelDims(), spec.ModelDims(hidden=128, heads=2, enc_layers=2, dec_layers=3, enc_ff=96, dec_ff=96, var_filter=32,
                                                  var_bins=16, mel=20, max_len=300)):
        names, shapes, offset, total = spec.arena_layout(dims)
        plan = dp.bucket_plan(dims)
        tags = [t for t, _ in plan]
        assert tags == ([f"dec{i}" for i in reversed(range(dims.dec_layers))] + [f"enc{i}" for i in reversed(range(dims.enc_layers))] + ["tail"])
        flat = sorted(r for _, rs in plan for r in rs)
        assert flat[0][0] == 0 and flat[-1][1] == total and all(a[1] == b[0] for a, b in zip(flat, flat[1:])), "ranges must tile the arena"
        where = {}
        for t, rs in plan:
            for b, e in rs:
                for n in names:
                    if b <= offset[n] < e:
                        where[n] = t
        H = dims.hidden
        assert where["decoder.layers.1.ff.linear1.weight"] == "dec1" and where["decoder.layers.1.self_attn.w_v.weight"] == "dec1"
        assert where["transformer_encoder_layers.0.self_attn.w_o.weight"] == "enc0"
        # not final at the end of their layer: the batched cross K/V projections, every small vector, embeddings, heads
        for n in ("decoder.layers.1.cross_attn.w_k.weight", "decoder.layers.1.norm2.weight", "decoder.layers.1.ff.linear1.bias",
                  "decoder.layers.0.self_attn.q_norm.weight", "text_embedding.weight", "mel_projection_out.weight",
                  "duration_adaptor.variance_adaptor.pitch_predictor.conv_layers.0.weight", "positional_encoding.pe"):
            assert where[n] == "tail", n
        big = sum(e - b for t, rs in plan if t != "tail" for b, e in rs)
        frac = big / sum(int(np.prod(s)) for n, s in spec.param_shapes(dims).items())
        assert frac > (0.85 if dims.hidden == 512 else 0.7), f"the layer buckets carry the bulk of the bytes ({frac:.2f})"
        # a decoder layer's q|k|v projections are adjacent: they travel as one range
        rs = dict(plan)["dec0"]
        assert any(e - b >= 3 * H * H for b, e in rs)


def _bucket_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from kokoro_ruslan_amd import dp, spec
    dp.init("gloo")
    dims = spec.ModelDims(hidden=128, heads=2, enc_layers=2, dec_layers=2, enc_ff=96, dec_ff=96, var_filter=32, var_bins=16, mel=20, max_len=300)
    total = spec.arena_layout(dims)[3]
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(total, generator=g)
    whole = flat.clone()
    dist.all_reduce(whole, op=dist.ReduceOp.SUM)                        # what one exchange of the whole arena gives
    ex = dp.BucketedExchange(dims, world, backend="dist")
    ex.begin_step()
    for tag in ex.plan:                                                  # the engine issues the buckets in this order
        ex.reduce(flat, tag)
    dp.barrier()
    # bench.py's guard: after the exchange every rank holds the same arena; one differing element on one rank is seen
    in_step = dp.replicas_in_step(flat)
    other = flat.clone()
    if rank == 1:
        other[total // 2] += 1e-3
    diverged_seen = not dp.replicas_in_step(other)
    # round 5: the buckets travel in GROUPS (one exchange per group, issued when its last bucket is final) — the engine's arrival
    # order may differ from the plan's (encoder buckets come from the side branch): any order must give the sum of the whole arena
    grouped_ok = True
    for n_groups, order in ((2, list(ex.plan)), (3, ["enc1", "dec1", "enc0", "dec0", "tail"]), (1, list(ex.plan)[::-1]), (13, list(ex.plan)),
                            ("2+tail", list(ex.plan)), (None, ["enc1", "dec1", "enc0", "dec0", "tail"])):
        g2 = torch.Generator().manual_seed(100 + rank)
        f2 = torch.randn(total, generator=g2)
        ex2 = dp.BucketedExchange(dims, world, backend="dist", groups=n_groups)
        ex2.begin_step()
        calls = 0
        for tag in order:
            tags = ex2.arrive(tag)
            if tags is not None:
                ex2.reduce_tags(f2, tags)
                calls += 1
        want_calls = min(n_groups, len(ex.plan)) if isinstance(n_groups, int) else 3      # (the default: two groups + the tail alone)
        if not isinstance(n_groups, int):
            grouped_ok = grouped_ok and ex2.groups[-1] == ["tail"] and len(ex2.groups) == 3 and "tail" not in ex2.groups[1]
        grouped_ok = grouped_ok and bool(torch.equal(f2, whole)) and calls == want_calls and sorted(ex2.issued) == sorted(ex.plan)
    if rank == 0:
        q.put((bool(torch.equal(flat, whole)) and grouped_ok, list(ex.issued), in_step, diverged_seen))
    dist.destroy_process_group()


def test_bucketed_exchange_two_ranks_equals_one_all_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, issued, in_step, diverged_seen = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same, "bucket by bucket must give exactly the sum of the whole arena"
    assert in_step and diverged_seen, "replicas_in_step must accept equal arenas and flag one differing element"
    assert issued == ["dec1", "dec0", "enc1", "enc0", "tail"]


# ---- the backend of the in-step exchange is agreed on collectively (a rank that cannot bind RCCL must not leave the
# others in a broadcast or with a communicator nobody else joined) -----------------------------------------------------------------
def _create_worker(rank, world, port, q, scenario):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from kokoro_ruslan_amd import dp, spec, lib as kk
    dp.init("gloo")

    class FakeLib:                                       # stands in for libkokoro_hip.so's kk_comm_* on a box without GPUs
        def __init__(self):
            self.calls, self.world = [], 0

        def kk_comm_load(self, path):
            return 1 if (scenario == "load_fails_on_rank1" and rank == 1) else 0

        def kk_comm_unique_id(self, buf):
            self.calls.append("id")
            return 1 if scenario == "id_fails_on_rank0" else 0

        def kk_comm_world(self):
            return self.world

        def kk_comm_init(self, r, w, uid):
            self.calls.append("init")
            if scenario == "init_fails_on_rank1" and rank == 1:
                return 1
            self.world = w
            return 0

        def kk_comm_destroy(self):
            self.calls.append("destroy")
            self.world = 0
            return 0

        def kk_last_error(self):
            return b"injected failure"
    fake = FakeLib()
    kk.load = lambda: fake
    dims = spec.ModelDims(hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=96, dec_ff=96, var_filter=32, var_bins=16, mel=20, max_len=300)
    ex = dp.BucketedExchange.create(dims, rank, world, torch.device("cpu"))
    # whatever was agreed on must work as a collective on every rank
    x = torch.ones(4) * (rank + 1)
    if ex.backend == "dist":
        dist.all_reduce(x)
    q.put((rank, ex.backend, list(fake.calls), fake.world, x.tolist()))
    dp.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario,backend", [("all_ok", "rccl"), ("load_fails_on_rank1", "dist"), ("id_fails_on_rank0", "dist"),
                                              ("init_fails_on_rank1", "dist")])
def test_exchange_backend_is_agreed_on_by_all_ranks(scenario, backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_create_worker, args=(r, 2, port, q, scenario)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [backend, backend], got                # never a mix
    for rank, be, calls, live, x in got:
        if backend == "rccl":
            assert "init" in calls and live == 2
        else:
            assert live == 0, "no rank may keep a communicator the others never joined"
            assert x == [3.0] * 4
        if scenario == "load_fails_on_rank1":
            assert "init" not in calls, "nobody enters ncclCommInitRank unless everybody will"
        if scenario == "init_fails_on_rank1" and rank == 0:
            assert calls[-1] == "destroy"


# ---- the fused-encoder failure check is a collective that EVERY rank enters (a rank whose own fused launch was already
# off returned before the all-reduce and left the others alone in it) -------------------------------------------------------------
def _encstack_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from kokoro_ruslan_amd import dp
    from kokoro_ruslan_amd.engine import KokoroEngine
    dp.init("gloo")

    class Stub:                                          # the attributes check_encoder_stack touches (no GPU engine on this box)
        device = torch.device("cpu")

        def __init__(self, fused, code):
            self.enc_fused, self._code = fused, code
            self._enc_sync, self._graphs = torch.tensor([code]), {"g": 1}

        def encoder_stack_error(self):
            return self._code
    # rank 0: fused launch on, a barrier timed out; rank 1: fused launch already off — it must still join and must raise too
    eng = Stub(True, 7) if rank == 0 else Stub(False, 0)
    raised = False
    try:
        KokoroEngine.check_encoder_stack(eng)
    except RuntimeError:
        raised = True
    # a second round where nothing is wrong: nobody raises, nobody hangs
    eng2 = Stub(True, 0) if rank == 0 else Stub(False, 0)
    KokoroEngine.check_encoder_stack(eng2)
    q.put((rank, raised, eng.enc_fused, len(eng._graphs)))
    dp.barrier()
    dist.destroy_process_group()


def test_encoder_stack_check_is_entered_by_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_encstack_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, True, False, 0), (1, True, False, 1)], got


# ---- round 6: the trainer's tripwire — a ONE-ULP difference in one parameter on one rank is caught, rank 0's
# state is re-broadcast, and the next check passes ------------------------------------------------------------------------------------
def _tripwire_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import types
    from kokoro_ruslan_amd import dp
    dp.init("gloo")
    g = torch.Generator().manual_seed(5)
    mk = lambda: torch.randn(70_001, generator=torch.Generator().manual_seed(5))
    arena = types.SimpleNamespace(p=mk(), m=mk() * 0.1, v=mk().abs(), ema=mk(), p16=None)
    eng = types.SimpleNamespace(arena=arena, opt_state=torch.zeros(16, dtype=torch.float64))
    first = dp.check_replicas(eng, where="start")                       # identical replicas: passes, nothing moves
    before = arena.p.clone()
    if rank == 1:                                                        # one ulp, one element, one rank (and a stale moment)
        i = 12_345
        arena.p[i] = torch.nextafter(arena.p[i], torch.tensor(float("inf")))
        arena.m[7] += 1.0
        eng.opt_state[3] = 9.0
    caught = not dp.check_replicas(eng, where="after the injected ulp")  # warns + re-broadcasts rank 0's slabs
    healed = dp.check_replicas(eng, where="after the re-broadcast")
    same = bool(torch.equal(arena.p, before)) and float(arena.m[7]) == float((mk() * 0.1)[7]) and float(eng.opt_state[3]) == 0.0
    # the float checksum the round-5 guard used cannot see a flipped low bit beside large values; the integer one must
    big = torch.full((1000,), 3.0e7)
    big2 = big.clone()
    if rank == 1:
        big2[500] = torch.nextafter(big2[500], torch.tensor(float("inf")))
    bit_seen = not dp.replicas_in_step(big2)
    nan_seen = not dp.replicas_in_step(torch.tensor([1.0, float("nan")]))
    out = torch.tensor([float(first), float(caught), float(healed), float(same), float(bit_seen), float(nan_seen)])
    dist.all_reduce(out, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(out.tolist())
    dist.destroy_process_group()


def test_replica_tripwire_catches_one_ulp_and_rebroadcasts():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tripwire_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    first, caught, healed, same, bit_seen, nan_seen = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert first == 1.0, "identical replicas must pass the tripwire"
    assert caught == 1.0, "a one-ulp difference in one parameter on one rank must be caught on EVERY rank"
    assert healed == 1.0 and same == 1.0, "after the re-broadcast every rank holds rank 0's parameters, moments and optimizer state"
    assert bit_seen == 1.0 and nan_seen == 1.0
