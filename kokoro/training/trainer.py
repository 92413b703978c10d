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
from typing import Dict, Optional

import torch

from kokoro.data.cached import (CachedFeatureDataset, FrameBudgetBatchSampler, collate_fn, length_based_batch_sampler,
                                split_indices, step_groups)
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


class KokoroTrainer:
    def __init__(self, config, vocab_size: int = 59):
        from kokoro_ruslan_amd.engine import KokoroEngine
        self.config = config
        self.rank, self.world, self.local = dp.init()
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local)
        n_all = len(CachedFeatureDataset(config.feature_cache_dir, memory_cache=False))
        tr_idx, va_idx = split_indices(n_all, config.validation_split)
        self.dataset = CachedFeatureDataset(config.feature_cache_dir, tr_idx, config.max_seq_length, config.use_memory_cache)
        self.val_dataset = CachedFeatureDataset(config.feature_cache_dir, va_idx, config.max_seq_length, config.use_memory_cache) if va_idx else None
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
            # real data = ragged shards: normalise every loss by the GLOBAL valid-element counts (dp.LossSync) instead of
            # pre-scaling per-rank means by 1/world, and feed the batch-shape heuristics the global-batch mel length
            self.engine.loss_sync = dp.LossSync(self.world)
            self.engine.dp_loss_scale = 1.0
        self.start_epoch, self.best_val, self.best_epoch = 0, float("inf"), -1
        logger.info("engine ready: %d params, %s math, %d train / %d val utterances, %d batches/epoch, world %d",
                    sum(math.prod(s) for s in self.engine.arena.shapes.values()), math_mode, len(self.dataset),
                    len(self.val_dataset) if self.val_dataset else 0, len(self.sampler), self.world)

    # ------------------------------------------------------------------
    def _to_device(self, batch):
        return {k: v.to(self.engine.device, non_blocking=True) for k, v in batch.items()}

    def train_epoch(self, epoch: int) -> float:
        cfg, G = self.config, max(1, self.config.gradient_accumulation_steps)
        self.sampler.epoch = epoch
        batches = self.sampler.batches()
        groups = step_groups(self.sampler.global_batches(), self.world) if self.world > 1 else None
        acc, losses, n = 0, torch.zeros(6, device=self.engine.device), 0
        for bi, idxs in enumerate(batches):
            batch = cap_batch(self._to_device(collate_fn([self.dataset[i] for i in idxs])))
            if groups is not None:      # longest (capped) mel length among this step's batches on all ranks
                self.engine.global_mel_length = min(2000, max(min(self.dataset.samples[i]["audio_length"], self.dataset.max_seq_length)
                                                              for g in groups[bi] for i in g))
            div = effective_accumulation_divisor(G, acc, bi, len(batches))
            boundary = (acc + 1 >= G) or (bi == len(batches) - 1)
            self.engine.micro_in_cycle = acc
            losses += self.engine.train_step(batch, div, boundary, self.sync if self.world > 1 else None)
            acc = 0 if boundary else acc + 1
            n += 1
        avg = (losses / max(n, 1)).cpu().tolist()          # the only host sync of the epoch
        if n == 0:
            logger.warning("epoch %d: no training batches", epoch + 1)
        logger.info("epoch %d train: total %.4f mel %.4f dur %.4f stop %.4f pitch %.4f energy %.4f", epoch + 1, *avg)
        return avg[0]

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
        saved_p, saved_sync = None, e.loss_sync
        e.loss_sync = None
        if e.arena.ema is not None:
            saved_p = e.arena.p.clone()
            e.arena.p.copy_(e.arena.ema)                     # evaluate the EMA replica
        acc = torch.zeros(10, device=e.device, dtype=torch.float64)      # 6 losses, sc sum, sc batches, f0 sum, f0 batches
        n = 0
        with e.fp32_math():                                  # validation runs without autocast (trainer.py:1821-1834)
            for idxs in self._val_batches():
                batch = cap_batch(self._to_device(collate_fn([self.val_dataset[j] for j in idxs])))
                out = e.forward_backward(batch, backward=False)
                acc[:6] += out["losses"].double()
                n += 1
                T = batch["mel_specs"].shape[1]
                valid = (torch.arange(T, device=e.device)[None, :] < batch["mel_lengths"][:, None])
                ok = batch["mel_lengths"] > 0
                m3 = valid[:, :, None].to(torch.float32)
                num = ((batch["mel_specs"] - out["mel"]) * m3).flatten(1).norm(dim=1)
                den = (batch["mel_specs"] * m3).flatten(1).norm(dim=1)
                sc_ok = ok & (den > 0)
                acc[6] += torch.where(sc_ok, num / den.clamp(min=1e-30), torch.zeros_like(num)).sum().double() / sc_ok.sum().clamp(min=1)
                acc[7] += (sc_ok.sum() > 0).double()
                se = ((batch["pitches"][:, :T] - out["pitch"]) ** 2 * valid).sum(1) / batch["mel_lengths"].clamp(min=1)
                acc[8] += torch.where(ok, se.sqrt(), torch.zeros_like(se)).sum().double() / ok.sum().clamp(min=1)
                acc[9] += (ok.sum() > 0).double()
        if saved_p is not None:
            e.arena.p.copy_(saved_p)                         # (the bf16 weight shadow was never touched)
        e.loss_sync = saved_sync
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
            logger.info("resumed from %s at epoch %d", path, self.start_epoch)
        patience = 0
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
                p = ckpt.save_checkpoint(self.engine, cfg, epoch, loss, cfg.output_dir, val, self.best_val, self.best_epoch)
                logger.info("checkpoint saved: %s", p)
            if val is not None and patience >= cfg.early_stopping_patience:
                logger.info("early stopping at epoch %d", epoch + 1)
                break
        if self.rank == 0:
            logger.info("final model saved: %s", ckpt.save_final_model(self.engine, cfg, cfg.output_dir))
