"""Host training loop around the MI355X engine, with the reference trainer's control flow for the hot path:
epoch loop → micro-batches with exact accumulation divisor (reference training/trainer.py:3344-3362) → optimizer
boundary (all decisions on the device, see kokoro_ruslan_amd/csrc/kk_optim.hip) → validation on the EMA weights
(trainer.py:1771-1985, forward + losses only) → checkpoints in the reference layout (kokoro.training.checkpoint).

Input data are the reference's cached features (kokoro.data.cached); the audio front-end / MFA / phonemizer stay on
the reference.  TensorBoard, profilers and MPS memory management of the reference trainer are out of scope.
"""
from __future__ import annotations

import logging
import math
import os
import queue
import threading
import time
from typing import Dict, List, Optional

import torch

from kokoro.data.cached import (CachedFeatureDataset, FrameBudgetBatchSampler, batch_layout, collate_fn, collate_into,
                                length_based_batch_sampler, scan_cache, split_indices, step_groups)
from kokoro.training import checkpoint as ckpt
from kokoro_ruslan_amd import dp, lib as kk
from kokoro_ruslan_amd.spec import ModelDims, StepHyper

logger = logging.getLogger(__name__)


def recommended_ema_decay(steps_per_epoch: int, half_life_epochs: float) -> float:
    """exp(-ln2 / (steps_per_epoch * k)) clipped to [0.9, 0.9999] (reference utils/ema.py:6-27)."""
    hl = steps_per_epoch * half_life_epochs
    if steps_per_epoch <= 0 or hl <= 0:
        return 0.9999
    return max(0.9, min(math.exp(-math.log(2) / hl), 0.9999))


def effective_accumulation_divisor(G: int, accumulated_step: int, batch_idx: int, num_batches: int) -> int:
    """trainer.py:3344-3362."""
    return max(1, min(max(1, int(G)), max(0, int(accumulated_step)) + max(1, int(num_batches) - int(batch_idx))))


def cap_batch(batch: Dict[str, torch.Tensor], max_mel: int = 2000, max_ph: int = 2000) -> Dict[str, torch.Tensor]:
    """_cap_batch_sequence_dimensions (trainer.py:3364-3411)."""
    b = dict(batch)
    if b["mel_specs"].size(1) > max_mel:
        for k in ("mel_specs", "stop_token_targets", "pitches", "energies"):
            b[k] = b[k][:, :max_mel].contiguous()
        b["mel_lengths"] = b["mel_lengths"].clamp(max=max_mel)
    if b["phoneme_indices"].size(1) > max_ph:
        for k in ("phoneme_indices", "phoneme_durations", "stress_indices"):
            b[k] = b[k][:, :max_ph].contiguous()
        b["phoneme_lengths"] = b["phoneme_lengths"].clamp(max=max_ph)
    return b


_TORCH_DT = {"float32": torch.float32, "int64": torch.int64}


class BatchPrefetcher:
    """Background loader for the train loop: while the GPU runs step n, a thread reads and collates batch n+1 into a
    PINNED staging slab and copies it to a device slab with ONE asynchronous H2D transfer on its own stream.

    The reference runs `DataLoader(num_workers=0)` with `pin_memory` off (cli/cli.py:275-276, trainer.py:322-327): every
    batch is loaded, collated and copied on the training thread, field by field, from pageable memory.  At a few
    milliseconds per step that serialises the step behind the loader, so here the batch is ready on the device when the
    step starts.  A ring of `depth` slab pairs; a slab is refilled only after the step that consumed it has finished on
    the GPU (an event recorded by the consumer), which is also what bounds the loader's run-ahead.

    Yields (batch of device views, expanded_len) where expanded_len = max_b sum(durations) is computed on the host
    copy, and the consuming stream already waits for the transfer."""

    def __init__(self, dataset, batches: List[List[int]], device: torch.device, depth: int = 3, max_mel: int = 2000,
                 max_ph: int = 2000, hip_lock=None):
        self.dataset, self.batches, self.device, self.depth = dataset, batches, device, max(2, depth)
        self.max_mel, self.max_ph = max_mel, max_ph      # _cap_batch_sequence_dimensions (reference trainer.py:3364-3411)
        self.hip_lock = hip_lock if hip_lock is not None else threading.Lock()    # engine.capture_lock: no HIP calls from
        #                                                                           this thread while a graph is captured
        self.device_index = device.index if device.index is not None else torch.cuda.current_device()
        self.copy_stream = torch.cuda.Stream(device=device)
        self.host = [None] * self.depth
        self.dev = [None] * self.depth
        # free slots, each with the event after which its slab may be overwritten (the step that read it is done); a slot
        # comes back only when the consumer hands it back, which is also what bounds the loader's run-ahead
        self.free_q: "queue.Queue" = queue.Queue()
        for slot in range(self.depth):
            self.free_q.put((slot, None))
        self.q: "queue.Queue" = queue.Queue()
        self.error: Optional[BaseException] = None
        self.load_s = 0.0                                # seconds this thread spent reading + collating + staging
        self.wait_s = 0.0                                # ... and waiting for the GPU to release a slab
        self.stop = False
        self.thread = threading.Thread(target=self._run, name="kokoro-batch-prefetch", daemon=True)
        self.thread.start()

    def _run(self) -> None:
        import numpy as np
        try:
            torch.cuda.set_device(self.device_index)
            host_np = [None] * self.depth
            for idxs in self.batches:
                if self.stop:
                    break
                t0 = time.perf_counter()
                items = [self.dataset[i] for i in idxs]
                B, T, P, M, mel_len, ph_len, plan, total = batch_layout(items, self.max_mel, self.max_ph)
                self.load_s += time.perf_counter() - t0
                slot, done = self.free_q.get()
                if slot is None:                         # consumer gone
                    break
                t0 = time.perf_counter()
                while done is not None and not done.query():      # the step that used this slab has left the GPU
                    time.sleep(0.0002)                           # (polled: a blocking wait would sit on the lock)
                self.wait_s += time.perf_counter() - t0
                t0 = time.perf_counter()
                if self.host[slot] is None or self.host[slot].numel() < total:
                    with self.hip_lock:
                        cap_bytes = total + total // 4
                        self.host[slot] = torch.empty(cap_bytes, dtype=torch.uint8).pin_memory()
                        self.dev[slot] = torch.empty(cap_bytes, dtype=torch.uint8, device=self.device)
                        torch.cuda.synchronize(self.device_index)   # (a recycled block may still be in use on another stream)
                    host_np[slot] = self.host[slot].numpy()
                host, dev, hnp = self.host[slot], self.dev[slot], host_np[slot]
                arrays, views = {}, {}
                for k, off, n, shape, dt in plan:
                    arrays[k] = hnp[off:off + n].view(dt).reshape(shape)
                    views[k] = dev[off:off + n].view(_TORCH_DT[dt]).view(shape)
                collate_into(items, arrays, mel_len, ph_len)
                expanded = int(np.clip(arrays["phoneme_durations"], 0, None).sum(axis=1).max()) if B and P else 0
                with self.hip_lock:
                    ready = torch.cuda.Event()
                    with torch.cuda.stream(self.copy_stream):
                        dev[:total].copy_(host[:total], non_blocking=True)
                        ready.record(self.copy_stream)
                self.load_s += time.perf_counter() - t0
                self.q.put((slot, views, expanded, ready))
            self.q.put(None)
        except BaseException as e:                       # surfaced on the training thread
            self.error = e
            self.q.put(None)

    def __iter__(self):
        try:
            while True:
                item = self.q.get()
                if item is None:
                    if self.error is not None:
                        raise self.error
                    return
                slot, views, expanded, ready = item
                torch.cuda.current_stream().wait_event(ready)
                yield views, expanded
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream())      # everything the consumer queued on this batch
                self.free_q.put((slot, done))
        finally:                                              # consumer left early: let the loader thread run out
            self.stop = True
            self.free_q.put((None, None))
            self.thread.join(timeout=30)


class KokoroTrainer:
    def __init__(self, config, vocab_size: int = 59):
        from kokoro_ruslan_amd.engine import KokoroEngine
        self.config = config
        self.rank, self.world, self.local = dp.init()
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local)
        metas = scan_cache(config.feature_cache_dir)          # one pass over the cache for both views (and an index file)
        tr_idx, va_idx = split_indices(len(metas), config.validation_split)
        self.dataset = CachedFeatureDataset(config.feature_cache_dir, tr_idx, config.max_seq_length, config.use_memory_cache, metas)
        self.val_dataset = (CachedFeatureDataset(config.feature_cache_dir, va_idx, config.max_seq_length, config.use_memory_cache, metas)
                            if va_idx else None)
        if config.use_dynamic_batching:
            self.sampler = FrameBudgetBatchSampler(self.dataset, config.max_frames_per_batch, config.min_batch_size,
                                                   config.max_batch_size, True, self.rank, self.world, drop_last=True)   # trainer.py:305-312
        else:
            self.sampler = length_based_batch_sampler(self.dataset, config.batch_size, True, self.rank, self.world,
                                                      drop_last=True)                                                  # trainer.py:315-320
        G = max(1, config.gradient_accumulation_steps)
        steps_per_epoch = max(1, -(-len(self.sampler) // G))
        hp = StepHyper.from_config(config)
        if config.ema_decay is None:
            hp.ema_decay = recommended_ema_decay(steps_per_epoch, config.ema_half_life_epochs)   # trainer.py:808-827
        dims = ModelDims(vocab=vocab_size, mel=config.n_mels, hidden=config.hidden_dim, heads=config.n_heads,
                         enc_layers=config.n_encoder_layers, dec_layers=config.n_decoder_layers, enc_ff=config.encoder_ff_dim,
                         dec_ff=config.decoder_ff_dim, var_filter=config.variance_filter_size,
                         var_kernel=config.variance_kernel_size, var_bins=config.n_variance_bins,
                         max_len=config.max_decoder_seq_len)
        math_mode = "bf16" if (config.use_mixed_precision and config.mixed_precision_dtype == "bfloat16") else "f32"
        self.engine = KokoroEngine(dims, hp, math_mode=math_mode, total_steps=config.num_epochs * steps_per_epoch, seed=0)
        self.sync = dp.GradSync(self.world)
        if self.world > 1:
            # real data = ragged shards: normalise every loss by the GLOBAL valid-element counts instead of pre-scaling
            # per-rank means by 1/world, and feed the batch-shape heuristics the global-batch mel length.
            # Both collectives of a step — the gradient buckets inside the backward and the loss normalisers between the loss
            # forward and the loss backward — go through ONE communicator behind the C ABI (kk_comm_*, RCCL) and are captured
            # with the step, so ragged data-parallel steps replay from hipGraphs like single-GPU ones.  KK_DP_LEGACY=1 (or
            # RCCL not bindable on some rank: dp.BucketedExchange.create agrees on that collectively) keeps the eager form:
            # torch.distributed all-reduces between the kernels / after the backward.
            self.engine.dp_loss_scale = 1.0
            comm = None if os.environ.get("KK_DP_LEGACY") == "1" else dp.BucketedExchange.create(dims, self.rank, self.world, self.engine.device)
            if comm is not None and comm.capturable:
                self.engine.dp_comm, self.engine.loss_sync, self.sync = comm, comm, None
            else:
                self.engine.loss_sync = dp.LossSync(self.world)
        self.start_epoch, self.best_val, self.best_epoch, self.patience = 0, float("inf"), -1, 0
        self.use_graphs = os.environ.get("KK_TRAINER_GRAPHS", "1") != "0"
        # Per-micro-batch non-finite guard (reference trainer.py:2304-2314): by default the device flags the micro-batch
        # and drops its accumulation cycle at the scheduled boundary, without any host round trip.  Strict mode also
        # RE-PHASES the cycle like the reference (the next batch starts a new cycle), which needs the flag on the host:
        # one synchronisation per micro-batch.
        self.strict_nonfinite_guard = os.environ.get("KK_STRICT_NONFINITE", "0") == "1"
        self.prefetch_depth = int(os.environ.get("KK_PREFETCH_DEPTH", "3"))
        # data parallel: replicas never exchange weights — identical reduced gradients through the identical (order-deterministic)
        # optimizer pass keep them bit-identical.  The tripwire checks that every `replica_check_every` optimizer steps
        # and at every epoch end (before validation / checkpoints), and re-broadcasts rank 0's state with a warning on divergence.
        self.replica_check_every = max(0, int(getattr(config, "replica_check_every", 500)))
        self.opt_steps = 0
        self.replica_resyncs = 0
        if not hp.use_onecycle_lr:
            from kokoro_ruslan_amd import spec as _spec
            _spec.cosine_restart_position(0, hp.lr_T_0, hp.lr_T_mult)      # (raises ValueError like torch's scheduler constructor)
        logger.info("engine ready: %d params, %s math, %d train / %d val utterances, %d batches/epoch, world %d",
                    sum(math.prod(s) for s in self.engine.arena.shapes.values()), math_mode, len(self.dataset),
                    len(self.val_dataset) if self.val_dataset else 0, len(self.sampler), self.world)

    # ------------------------------------------------------------------
    def _to_device(self, batch):
        return {k: v.to(self.engine.device, non_blocking=True) for k, v in batch.items()}

    def train_epoch(self, epoch: int) -> float:
        cfg, G, e = self.config, max(1, self.config.gradient_accumulation_steps), self.engine
        self.sampler.epoch = epoch
        self.engine.lr_epoch = epoch                 # legacy schedule (use_onecycle_lr = False): one scheduler step per completed epoch
        batches = self.sampler.batches()
        groups = step_groups(self.sampler.global_batches(), self.world) if self.world > 1 else None
        # model.train() with the configured regularisation (reference trainer.py:2038-2056): dropout, stochastic depth,
        # and SpecAugment on the decoder memory from spec_augment_start_epoch on
        e.train_dropout = True
        e.spec_augment_active = bool(cfg.use_spec_augment) and epoch >= int(cfg.spec_augment_start_epoch)
        step = e.train_step_auto if self.use_graphs else e.train_step
        acc, losses, n = 0, torch.zeros(6, device=e.device), 0
        try:
            self.last_prefetch = BatchPrefetcher(self.dataset, batches, e.device, self.prefetch_depth, 2000, 2000, e.capture_lock)
            for bi, (batch, expanded) in enumerate(self.last_prefetch):
                if groups is not None:      # longest (capped) mel length among this step's batches on all ranks
                    e.global_mel_length = min(2000, max(self.dataset.samples[i]["audio_length"] for g in groups[bi] for i in g))
                div = effective_accumulation_divisor(G, acc, bi, len(batches))
                boundary = (acc + 1 >= G) or (bi == len(batches) - 1)
                e.micro_in_cycle = acc
                T = batch["mel_specs"].shape[1]
                if self.strict_nonfinite_guard:
                    flag = e.opt_state[kk.OS["MICRO_BAD"]]
                    if acc == 0:
                        e.zero_grad()
                    e._exchange_now = boundary                # (in-step bucket exchange: only the boundary micro-batch communicates)
                    out = e.forward_backward(batch, loss_scale=e.dp_loss_scale / div, adaptive=True,
                                             expanded_len=expanded if expanded != T else None)["losses"]
                    e._exchange_now = True
                    if float(flag) != 0.0:                # (host sync) reset accumulation and skip, trainer.py:2304-2314
                        logger.error("batch %d: non-finite outputs or losses - accumulation reset, batch skipped", bi)
                        flag.zero_()
                        e.zero_grad()
                        acc = 0
                        continue
                    losses += out
                    if boundary:
                        if self.world > 1 and self.sync is not None:
                            self.sync(e.arena.g)
                        e.optimizer_step(int(e.global_mel_length or T))
                        e.micro_in_cycle = 0
                else:
                    losses += step(batch, div, boundary, self.sync if (self.world > 1 and self.sync is not None) else None, expanded if expanded != T else None)
                acc = 0 if boundary else acc + 1
                n += 1
                if boundary:
                    self.opt_steps += 1                       # (equal on every rank: the samplers hand out equal step counts)
                    if self.world > 1 and self.replica_check_every and self.opt_steps % self.replica_check_every == 0:
                        self._check_replicas(f"optimizer step {self.opt_steps}")
        finally:
            e.train_dropout = False
        avg = (losses / max(n, 1)).cpu().tolist()          # the only host sync of the epoch
        e.check_encoder_stack()                            # (a second word read at the same sync point: raises on a timed-out barrier)
        if self.world > 1:                                 # every rank, before anything (validation, checkpoint) reads the weights
            self._check_replicas(f"end of epoch {epoch + 1}")
        if n == 0:
            logger.warning("epoch %d: no training batches", epoch + 1)
        logger.info("epoch %d train: total %.4f mel %.4f dur %.4f stop %.4f pitch %.4f energy %.4f", epoch + 1, *avg)
        return avg[0]

    def _check_replicas(self, where: str) -> bool:
        ok = dp.check_replicas(self.engine, logger, where)
        if not ok:
            self.replica_resyncs += 1
        return ok

    def _val_batches(self):
        """Validation batches like the reference (trainer.py:331-348): same sampler family, no shuffle, nothing dropped;
        every rank validates the whole split (no gradient exchange happens here)."""
        cfg = self.config
        if cfg.use_dynamic_batching:
            s = FrameBudgetBatchSampler(self.val_dataset, cfg.max_frames_per_batch, cfg.min_batch_size, cfg.max_batch_size, False)
        else:
            s = length_based_batch_sampler(self.val_dataset, cfg.batch_size, False)
        return s.global_batches()

    @torch.no_grad()
    def validate_epoch(self) -> Optional[Dict[str, float]]:
        """Losses on the EMA weights in fp32 + the reference's two validation metrics (trainer.py:1866-1910): spectral
        convergence ||ref - pred||_F / ||ref||_F and frame-level F0 RMSE, each averaged per sample, then per batch, then
        over batches — computed on the device with masks, one host read at the end of the epoch."""
        if not self.val_dataset or len(self.val_dataset) == 0:
            return None
        e = self.engine
        e.check_encoder_stack()
        acc = torch.zeros(10, device=e.device, dtype=torch.float64)      # 6 losses, sc sum, sc batches, f0 sum, f0 batches
        n = 0
        saved_sync, saved_drop, e.loss_sync, e.train_dropout = e.loss_sync, e.train_dropout, None, False
        try:
            # validation runs without autocast on the EMA replica (trainer.py:1821-1834, 1771-1790): fp32 arithmetic,
            # and the EMA slab read in place (no copy; nothing to restore but three references, whatever happens)
            with e.fp32_math(), e.ema_weights():
                for idxs in self._val_batches():
                    cpu = cap_batch(collate_fn([self.val_dataset[j] for j in idxs]))
                    expanded = int(cpu["phoneme_durations"].clamp(min=0).sum(dim=1).max())
                    batch = self._to_device(cpu)
                    T = batch["mel_specs"].shape[1]
                    out = e.forward_backward(batch, backward=False, expanded_len=expanded if expanded != T else None)
                    acc[:6] += out["losses"].double()
                    n += 1
                    valid = (torch.arange(T, device=e.device)[None, :] < batch["mel_lengths"][:, None])
                    ok = batch["mel_lengths"] > 0
                    m3 = valid[:, :, None].to(torch.float32)
                    num = ((batch["mel_specs"] - out["mel"]) * m3).flatten(1).norm(dim=1)
                    den = (batch["mel_specs"] * m3).flatten(1).norm(dim=1)
                    sc_ok = ok & (den > 0)
                    acc[6] += torch.where(sc_ok, num / den.clamp(min=1e-30), torch.zeros_like(num)).sum().double() / sc_ok.sum().clamp(min=1)
                    acc[7] += (sc_ok.sum() > 0).double()
                    se = ((batch["pitches"][:, :T] - out["pitch"][:, :T]) ** 2 * valid).sum(1) / batch["mel_lengths"].clamp(min=1)
                    acc[8] += torch.where(ok, se.sqrt(), torch.zeros_like(se)).sum().double() / ok.sum().clamp(min=1)
                    acc[9] += (ok.sum() > 0).double()
        finally:
            e.loss_sync, e.train_dropout = saved_sync, saved_drop
        v = acc.cpu().tolist()
        res = dict(zip(("total", "mel", "dur", "stop", "pitch", "energy"), (x / max(n, 1) for x in v[:6])))
        res["spectral_convergence"] = v[6] / v[7] if v[7] > 0 else None
        res["f0_rmse"] = v[8] / v[9] if v[9] > 0 else None
        return res

    def train(self) -> None:
        cfg = self.config
        os.makedirs(cfg.output_dir, exist_ok=True)
        resume = cfg.resume_checkpoint
        path = ckpt.find_latest_checkpoint(cfg.output_dir) if resume == "auto" else resume
        if path and os.path.exists(path):
            c = ckpt.load_checkpoint(self.engine, path)
            self.start_epoch = int(c["epoch"]) + 1
            self.best_val = c.get("best_val_loss") or float("inf")
            self.best_epoch = c.get("best_val_epoch", -1)
            self.patience = int(c.get("early_stopping_counter", 0))
            logger.info("resumed from %s at epoch %d", path, self.start_epoch)
        patience = self.patience
        for epoch in range(self.start_epoch, cfg.num_epochs):
            loss = self.train_epoch(epoch)
            val = self.validate_epoch() if (epoch + 1) % max(1, cfg.validation_interval) == 0 else None
            improved = False
            if val is not None:
                logger.info("epoch %d val: total %.4f mel %.4f dur %.4f stop %.4f", epoch + 1, val["total"], val["mel"], val["dur"], val["stop"])
                if val.get("spectral_convergence") is not None:
                    logger.info("  SpectralConv: %.6f  f0_RMSE: %.6f", val["spectral_convergence"], val["f0_rmse"] or 0.0)
                if val["total"] < self.best_val - cfg.early_stopping_min_delta:
                    self.best_val, self.best_epoch, improved, patience = val["total"], epoch, True, 0
                else:
                    patience += 1
            periodic = (epoch + 1) % max(1, cfg.save_every) == 0 and (not self.val_dataset or patience > 0)     # trainer.py:2986-2998
            if self.rank == 0 and (improved or periodic or epoch + 1 == cfg.num_epochs):
                p = ckpt.save_checkpoint(self.engine, cfg, epoch, loss, cfg.output_dir, val, self.best_val, self.best_epoch, patience)
                logger.info("checkpoint saved: %s", p)
            if val is not None and patience >= cfg.early_stopping_patience:
                logger.info("early stopping at epoch %d", epoch + 1)
                break
        if self.rank == 0:
            logger.info("final model saved: %s", ckpt.save_final_model(self.engine, cfg, cfg.output_dir))
