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

import os
from typing import Dict, Optional, Tuple

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

    # Overlapped form (engine.train_step_graphed): `start` launches the exchange of the ranges that are already final
    # asynchronously — RCCL runs on its own stream after the work queued so far, the caller keeps queueing the rest of
    # the backward — and `finish` exchanges the remaining ranges and makes the current stream wait for everything.
    def start(self, flat_grad: torch.Tensor, ranges):
        works = []
        if self._active():
            for beg, end in ranges:
                for o in range(beg, end, self.bucket):
                    works.append(dist.all_reduce(flat_grad[o:min(end, o + self.bucket)], op=dist.ReduceOp.SUM, async_op=True))
        return works

    def finish(self, flat_grad: torch.Tensor, ranges, works) -> None:
        if not self._active():
            return
        for beg, end in ranges:
            for o in range(beg, end, self.bucket):
                dist.all_reduce(flat_grad[o:min(end, o + self.bucket)], op=dist.ReduceOp.SUM)
        for w in works:
            w.wait()


class LossSync:
    """engine.loss_sync hook: global loss normalisers for ragged data-parallel shards (collective (2))."""

    def __init__(self, world: int):
        self.world = world

    def __call__(self, acc: torch.Tensor, max_dur: torch.Tensor) -> None:
        if self.world == 1:
            return
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)        # 5 numerators, 5 valid-element counts (fp64)
        dist.all_reduce(max_dur, op=dist.ReduceOp.MAX)    # adaptive loss scale / clip heuristics: same inputs on every rank


def all_max(x: torch.Tensor) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return x


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown() -> None:
    """Tear the process group down (quietens the "destroy_process_group() was not called" warning at exit)."""
    if dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:
            pass
