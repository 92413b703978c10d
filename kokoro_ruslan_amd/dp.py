"""Data-parallel exchange for the Kokoro train step: one process per GPU, gradients summed over RCCL (xGMI).

The reference has no distributed code; the contract here is "same maths as one process seeing
the global batch".  Each rank runs forward+backward on its shard with the loss scaled by 1/world, the flat
gradient arena is SUM-all-reduced, and every rank then runs the identical optimizer pass on the identical reduced
gradients (pre-clip, norm, clip, AdamW, EMA are deterministic functions of them), so replicas stay in step without
any further collective.  That is exact when every rank holds the same number of valid loss elements (the fixed-shape
configurations, bench.py).  For ragged shards `LossSync` additionally SUM-reduces the 5 loss sums + 5 valid-element
counts and MAX-reduces the batch's largest duration between the loss forward and the loss backward (80 + 8 bytes):
every rank then normalises by the GLOBAL counts (and the loss is not pre-scaled by 1/world), so the summed gradients
are the global-batch gradients, and the batch-shape heuristics of trainer.py:2218-2242 see the same inputs everywhere.

The functions take plain tensors so the same code runs under `gloo` on CPU (tests) and `nccl` (= RCCL) on MI355X.
"""
from __future__ import annotations

import logging
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when not launched distributed."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, world, local = env_world()
    if (world > 1 or os.environ.get("KK_DP_FORCE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Rank r takes samples [r*B/world, (r+1)*B/world) of a global batch (SURVEY §8e partitioning)."""
    B = batch["mel_specs"].shape[0]
    if B % world:
        raise ValueError(f"global batch {B} not divisible by world size {world}")
    per = B // world
    return {k: v[rank * per:(rank + 1) * per].contiguous() for k, v in batch.items()}


class GradSync:
    """SUM all-reduce of the flat gradient arena in fixed-size buckets (one large message per bucket keeps the
    ring per-link bound instead of latency bound; 7 x ~153 GB/s xGMI links per GPU)."""

    def __init__(self, world: int, bucket_elems: int = 32 * 1024 * 1024, force: bool = False):
        self.world = world
        self.bucket = bucket_elems
        self.force = force          # exercise the collectives in a 1-rank group too (tests / KK_DP_FORCE=1)

    @property
    def loss_scale(self) -> float:
        """Fold the 1/world of the gradient mean into the loss so the collective is a plain SUM."""
        return 1.0 / self.world

    def _active(self) -> bool:
        return self.world > 1 or (self.force and dist.is_initialized())

    def __call__(self, flat_grad: torch.Tensor) -> None:
        if not self._active():
            return
        n = flat_grad.numel()
        for o in range(0, n, self.bucket):
            dist.all_reduce(flat_grad[o:min(n, o + self.bucket)], op=dist.ReduceOp.SUM)


class LossSync:
    """engine.loss_sync over torch.distributed: global loss normalisers for ragged data-parallel shards (SURVEY §8e,
    collective (2)).  Eager only (capturable = False) — the step's own communicator carries the same two collectives inside the
    hipGraph: BucketedExchange.loss_sync over RCCL."""
    capturable = False

    def __init__(self, world: int):
        self.world = world

    def loss_sync(self, acc: torch.Tensor, max_dur: torch.Tensor) -> None:
        if self.world == 1:
            return
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)        # 5 numerators, 5 valid-element counts, non-finite count (fp64)
        dist.all_reduce(max_dur, op=dist.ReduceOp.MAX)    # adaptive loss scale / clip heuristics: same inputs on every rank

    __call__ = loss_sync


# ---------------------------------------------------------------------------------------------------------------------
# Bucketed exchange inside the step: buckets in the order the backward finishes them, each all-reduced on
# a communication stream while the rest of the backward runs — captured into the step's hipGraph as one more branch.
# ---------------------------------------------------------------------------------------------------------------------
def bucket_plan(dims) -> "List[Tuple[str, List[Tuple[int, int]]]]":
    """[(tag, [(begin, end) element ranges of the gradient arena])] in issue order.

    A layer's weight-gradient GEMMs run as one grouped launch at the end of its backward, so its weight MATRICES are final
    there (98 % of the bytes): bucket "dec{i}" / "enc{i}" = those matrices, issued from decoder layer L-1 down to 0 and from
    encoder layer L-1 down to 0 (reverse-autograd order).  Everything else — the small vectors (their column sums are
    reduced once, at the end of the backward), the batched cross-attention K/V projections, embeddings, predictors, heads —
    is the "tail" bucket after the backward's last launch.  The buckets partition [0, total) (segment padding included)."""
    from . import spec
    names, shapes, offset, total = spec.arena_layout(dims)
    size = {n: -(-_prod(shapes[n]) // spec.SEG_ALIGN) * spec.SEG_ALIGN for n in names}
    mats = ("self_attn.w_q.weight", "self_attn.w_k.weight", "self_attn.w_v.weight", "self_attn.w_o.weight",
            "cross_attn.w_q.weight", "cross_attn.w_o.weight", "ff.linear1.weight", "ff.linear2.weight")
    tag_of = {}
    for i in range(dims.dec_layers):
        for m in mats:
            tag_of[f"decoder.layers.{i}.{m}"] = f"dec{i}"
    for i in range(dims.enc_layers):
        for m in mats:
            if not m.startswith("cross_attn"):
                tag_of[f"transformer_encoder_layers.{i}.{m}"] = f"enc{i}"
    ranges: Dict[str, List[List[int]]] = {}
    for n in names:                                   # physical order: adjacent segments of a bucket merge
        r = ranges.setdefault(tag_of.get(n, "tail"), [])
        if r and r[-1][1] == offset[n]:
            r[-1][1] = offset[n] + size[n]
        else:
            r.append([offset[n], offset[n] + size[n]])
    order = [f"dec{i}" for i in reversed(range(dims.dec_layers))] + [f"enc{i}" for i in reversed(range(dims.enc_layers))] + ["tail"]
    return [(t, [tuple(x) for x in ranges[t]]) for t in order if t in ranges]


def group_buckets(plan, n_groups: int) -> "List[List[str]]":
    """The plan's buckets (issue order) as n_groups consecutive groups of about equal bytes; the last bucket ("tail": final only when
    the whole backward is) closes the last group.  Deterministic in (plan, n_groups): every rank forms the same groups."""
    tags = list(plan.keys()) if hasattr(plan, "keys") else [t for t, _ in plan]
    size = {t: sum(e - b for b, e in (plan[t] if hasattr(plan, "keys") else dict(plan)[t])) for t in tags}
    n_groups = min(n_groups, len(tags))
    total, groups, acc = sum(size.values()), [[]], 0
    for i, t in enumerate(tags):
        groups[-1].append(t)
        acc += size[t]
        left, need = len(tags) - 1 - i, n_groups - len(groups)        # buckets left / groups still to open
        if need > 0 and left >= need and (acc >= total * len(groups) / n_groups or left == need):
            groups.append([])
    return [g for g in groups if g]


def _prod(shape) -> int:
    n = 1
    for v in shape:
        n *= int(v)
    return n


class BucketedExchange:
    """engine.dp_comm: SUM-reduces the gradient arena bucket by bucket as the backward releases them.

    backend "rccl": the C ABI's kk_comm_* (RCCL called directly, on `stream`, legal inside a hipGraph capture);
    backend "dist": torch.distributed all-reduces over the same ranges (gloo on CPU for tests, or when RCCL cannot be
    bound).  payload "bf16" narrows each range into a bf16 staging arena before the exchange and widens it back
    (half the xGMI bytes; sums of bf16-rounded gradients)."""

    def __init__(self, dims, world: int, backend: str = "dist", payload: str = "f32", stream=None, groups: Optional[int] = None):
        self.plan = OrderedDict(bucket_plan(dims))
        self.world, self.backend, self.payload, self.stream = world, backend, payload, stream
        self.issued: List[str] = []
        self._tables: Dict[str, tuple] = {}
        self._g16: Optional[torch.Tensor] = None
        # Exchange GROUPS: consecutive buckets of the plan that travel as ONE exchange, issued when the last of them is final.  On this
        # runtime a kernel on the communication branch of the step's hipGraph costs the step +0.85 ms as soon as its group is closed by a
        # bucket of the MAIN chain (a decoder layer) — whatever the kernel, one group or thirteen — and +0.08 ms when the group is closed
        # on the side branch or by the tail (profiles/r05_dp_exchange_one_gpu_ab.txt; a one-rank fp32 all-reduce enqueues no kernel, which
        # hid this until round 5).  Default: TWO groups of about equal bytes, {dec5..dec0, enc5} and {enc4..enc0, tail} at default dims
        # (KK_DP_GROUPS = a count, or "dec3,enc0" = explicit closing buckets for probes).
        # Round 5, late: the default is "2+tail" — the second of the two groups gives up the tail bucket to a third: {enc4..enc0} then
        # closes with the encoder's backward (260 us before the end of the backward at 8x512) and only the tail (small vectors, the
        # batched cross K/V projections, embeddings, predictors, heads: ~10 % of the bytes) waits for the last launch.  One-GPU price of
        # the third group: +0.02 ... +0.04 ms with a kernel on the branch (profiles/r05_dp_exchange_one_gpu_ab.txt); what it saves in
        # exposed exchange at N > 1 is an estimate until a multi-GPU box measures it.
        g = groups if groups is not None else os.environ.get("KK_DP_GROUPS", "2+tail")
        if isinstance(g, str) and g.strip() == "2+tail":
            self.groups = group_buckets(self.plan, 2)
            if len(self.groups[-1]) > 1:
                self.groups = self.groups[:-1] + [self.groups[-1][:-1], self.groups[-1][-1:]]
        elif isinstance(g, str) and not g.strip().isdigit():    # "dec3,dec0": explicit groups, each closed by the named bucket (probes)
            ends, self.groups = set(g.split(",")), [[]]
            for t in self.plan:
                self.groups[-1].append(t)
                if t in ends and t != list(self.plan)[-1]:
                    self.groups.append([])
        else:
            self.groups = group_buckets(self.plan, max(1, int(g)))
        self._group_of = {t: gi for gi, tags in enumerate(self.groups) for t in tags}
        self._seen: Dict[int, List[str]] = {}

    def begin_step(self) -> None:
        self.issued = []
        self._seen = {}

    def arrive(self, tag: str) -> Optional[List[str]]:
        """Bucket `tag` is final.  Returns the tags of its group when the group is complete (exchange them now, in one call of
        reduce_tags), else None.  Host-side bookkeeping in issue order: identical on every rank."""
        gi = self._group_of[tag]
        seen = self._seen.setdefault(gi, [])
        seen.append(tag)
        return list(self.groups[gi]) if len(seen) == len(self.groups[gi]) else None

    @property
    def capturable(self) -> bool:
        """True when every collective of this exchange may be captured into the step's hipGraph (RCCL through the C ABI)."""
        return self.backend == "rccl"

    def loss_sync(self, acc: torch.Tensor, max_dur: torch.Tensor) -> None:
        """engine.loss_sync: SUM of the loss sums / valid-element counts, MAX of the largest duration, in place, on the CURRENT
        stream (they sit on the step's critical path between kk_losses_fwd and kk_losses_finalize) and through the SAME
        communicator as the gradient buckets — one communicator per step, nothing a hipGraph cannot hold."""
        if self.backend == "dist":
            if dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(acc, op=dist.ReduceOp.SUM)
                dist.all_reduce(max_dur, op=dist.ReduceOp.MAX)
            return
        from . import lib as kk
        kk.call("kk_comm_loss_sync", acc, acc.numel(), max_dur)

    def reduce_tags(self, flat: torch.Tensor, tags: List[str]) -> None:
        """Exchange the buckets `tags` as ONE exchange (one RCCL group, one cast launch per direction)."""
        key = "+".join(tags)
        if key not in self.plan:
            merged: List[List[int]] = []
            for b, e in sorted(r for t in tags for r in self.plan[t]):
                if merged and merged[-1][1] == b:
                    merged[-1][1] = e
                else:
                    merged.append([b, e])
            self.plan_merged = getattr(self, "plan_merged", {})
            self.plan_merged[key] = [tuple(x) for x in merged]
        self.issued.extend(tags[:-1])
        self.reduce(flat, tags[-1], _ranges=self.plan_merged[key] if key not in self.plan else self.plan[key], _key=key)

    def reduce(self, flat: torch.Tensor, tag: str, _ranges=None, _key=None) -> None:
        """Exchange bucket `tag` of the flat gradient arena in place (on the current stream)."""
        ranges = self.plan[tag] if _ranges is None else _ranges
        self.issued.append(tag)
        tag = tag if _key is None else _key
        if self.backend == "dist":
            for b, e in ranges:
                dist.all_reduce(flat[b:e], op=dist.ReduceOp.SUM)
            return
        import ctypes as C
        from . import lib as kk
        if tag not in self._tables:
            n = len(ranges)
            self._tables[tag] = ((C.c_int64 * n)(*[b for b, _ in ranges]), (C.c_int64 * n)(*[e for _, e in ranges]), n)
        beg, end, n = self._tables[tag]
        if self.payload == "bf16":
            if self._g16 is None or self._g16.numel() != flat.numel():
                self._g16 = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            # (one thin launch per direction and bucket: kk_cast_ranges — the per-range casts with chip-filling grids cost the chain
            #  they ran beside 42 % on one GPU)
            kk.call("kk_cast_ranges", flat, self._g16, beg, end, n, 1, 1.0)
            kk.call("kk_comm_reduce_ranges", self._g16, beg, end, n, 1)
            kk.call("kk_cast_ranges", self._g16, flat, beg, end, n, 0, 1.0)
        else:
            kk.call("kk_comm_reduce_ranges", flat, beg, end, n, 0)
            if os.environ.get("KK_DP_PROBE_KERNEL"):           # probe: a one-rank fp32 exchange enqueues NO kernel; this stands in for a collective's
                if getattr(self, "_probe_slot", None) is None:
                    self._probe_slot = torch.zeros(2, dtype=torch.int64, device=flat.device)
                kk.call("kk_timestamp", self._probe_slot)

    @classmethod
    def create(cls, dims, rank: int, world: int, device: torch.device, payload: Optional[str] = None) -> "BucketedExchange":
        """Communicator over RCCL through the C ABI: rank 0 draws the id, the torch.distributed group (already up for
        the rendezvous) carries it to the others.  The backend is agreed on COLLECTIVELY: every rank takes part in the
        broadcast of the id whatever happened to it locally, then one MIN all-reduce of an "RCCL is up here" flag decides;
        if any rank failed, the ranks that succeeded destroy their communicator and ALL ranks use the "dist" backend
        (the same ranges over torch.distributed) — never a mix, which would hang or mismatch collectives."""
        import ctypes as C
        import logging
        from . import lib as kk
        log = logging.getLogger(__name__)
        payload = payload or os.environ.get("KK_DP_PAYLOAD", "f32")
        lib = kk.load()
        multi = world > 1 and dist.is_initialized()
        err = None
        buf = (C.c_char * 128)()
        try:                                              # stage 1 (local): bind RCCL, rank 0 draws the id
            import torch as _t
            path = os.path.join(os.path.dirname(_t.__file__), "lib", "librccl.so")
            if lib.kk_comm_load(path.encode() if os.path.exists(path) else None) != 0:
                raise RuntimeError(lib.kk_last_error().decode())
            if rank == 0 and lib.kk_comm_unique_id(buf) != 0:
                raise RuntimeError(lib.kk_last_error().decode())
        except Exception as e:
            err = e
        box = [bytes(buf) if (rank == 0 and err is None) else None]
        if multi:
            dist.broadcast_object_list(box, src=0)         # every rank, always (None = rank 0 could not draw an id)
        if err is None and box[0] is None:
            err = RuntimeError("rank 0 could not create the RCCL id")
        if multi:                                         # nobody enters ncclCommInitRank unless everybody will
            err = cls._agree(err, device, "kk_comm unavailable on some rank (%s)", log)
        if err is None:
            try:                                          # stage 2: the communicator itself (collective inside RCCL)
                if lib.kk_comm_world() == 0 and lib.kk_comm_init(rank, world, box[0]) != 0:
                    raise RuntimeError(lib.kk_last_error().decode())
            except Exception as e:
                err = e
            if multi:
                err = cls._agree(err, device, "kk_comm_init failed on some rank (%s)", log)
            if err is not None:
                lib.kk_comm_destroy()                      # (no-op where no communicator exists)
        stream = torch.cuda.Stream(device=device) if torch.cuda.is_available() else None
        if err is None:
            return cls(dims, world, "rccl", payload, stream)
        log.warning("kk_comm unavailable (%s); gradient buckets go through torch.distributed on every rank", err)
        return cls(dims, world, "dist", "f32", stream)

    @staticmethod
    def _agree(err, device, fmt, log):
        """MIN all-reduce of a per-rank success flag: returns `err` unchanged when every rank succeeded or this rank failed,
        and an error for the ranks that succeeded while another one did not."""
        ok = torch.tensor([0.0 if err is not None else 1.0], device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok) < 1.0 and err is None:
            err = RuntimeError(fmt % "another rank")
        return err


def all_max(x: torch.Tensor) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return x


def _arena_checksum(flat: torch.Tensor) -> torch.Tensor:
    """Two exact int64 checksums of a flat fp32 tensor's BITS (sum of the words, sum of the words weighted by a position pattern):
    integer arithmetic, so a single flipped mantissa bit anywhere changes them, whatever the magnitudes around it."""
    w = flat.contiguous().view(torch.int32).to(torch.int64)
    pos = (torch.arange(w.numel(), device=w.device, dtype=torch.int64) % 8191) + 1
    return torch.stack([w.sum(), (w * pos).sum()])


def replicas_in_step(flat_params: torch.Tensor) -> bool:
    """True when every rank holds bit-identical parameters (data-parallel replicas never exchange weights: identical reduced
    gradients through the identical optimizer pass keep them equal, so any difference means the exchange or the pass went wrong).
    Exact integer checksums of the bits, MIN- and MAX-reduced; non-finite parameters also fail.  Collective — call it on every rank."""
    finite = bool(torch.isfinite(flat_params.double().sum()))
    chk = _arena_checksum(flat_params)
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return finite
    if dist.get_backend() != "nccl":
        chk = chk.cpu()
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    ok = torch.tensor([1.0 if finite else 0.0], device=lo.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return bool(float(ok) == 1.0 and torch.equal(lo, hi))


def resync_replicas(slabs, src: int = 0) -> None:
    """Re-broadcast rank `src`'s copy of every tensor in `slabs` (parameters, Adam moments, EMA, bf16 shadow): what the trainer does
    when replicas_in_step() fails ("verify with a periodic parameter-checksum all-reduce").  Collective."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    for t in slabs:
        if t is None:
            continue
        if dist.get_backend() == "nccl" or not t.is_cuda:
            dist.broadcast(t, src=src)
        else:                                             # (gloo with device tensors: through the host)
            h = t.cpu()
            dist.broadcast(h, src=src)
            t.copy_(h)


def check_replicas(engine, log=None, where: str = "") -> bool:
    """The trainer's tripwire: True when the replicas agree; otherwise warns, re-broadcasts rank 0's state (parameters, moments, EMA,
    bf16 weight shadow) and returns False.  Collective."""
    a = engine.arena
    if replicas_in_step(a.p):
        return True
    (log or logging.getLogger(__name__)).warning(
        "data-parallel replicas diverged%s: re-broadcasting rank 0's parameters, Adam moments and EMA", f" ({where})" if where else "")
    resync_replicas([a.p, a.m, a.v, a.ema, a.p16, engine.opt_state])
    return False


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown() -> None:
    """Tear the communicator and the process group down (quietens the "destroy_process_group() was not called" warning)."""
    try:
        from . import lib as kk
        if kk._lib is not None:
            kk._lib.kk_comm_destroy()
    except Exception:
        pass
    if dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:
            pass
