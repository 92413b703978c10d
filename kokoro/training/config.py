"""`TrainingConfig` with the reference's field names and defaults (reference: training/config.py:16-352).

Checkpoints pickle this object under the path ``kokoro.training.config.TrainingConfig`` (reference
training/trainer.py:47, checkpoint_manager.py:528-544), so the class lives at exactly that path.  The dataclass is
generated from the table below; `__post_init__` keeps the two side effects that matter off-MPS
(config.py:358-360 feature_cache_dir default, :384-386 checkpoint_segments >= 1).  Fields after the marker are new
(MI355X engine knobs) and default to the reference's behaviour.
"""
from __future__ import annotations

import dataclasses
from pathlib import Path
from typing import Optional

import torch

_DEVICE = "cuda" if torch.cuda.is_available() else "cpu"

# (name, type, default) — order as in the reference dataclass
_FIELDS = [
    ("data_dir", str, "data/processed_data"), ("output_dir", str, "output_models"), ("num_epochs", int, 30),
    ("batch_size", int, 16), ("learning_rate", float, 5.0e-5), ("device", str, _DEVICE),
    ("gradient_accumulation_steps", int, 2),
    ("use_onecycle_lr", bool, True), ("max_lr_multiplier", float, 1.0), ("pct_start", float, 0.20),
    ("encoder_lr_multiplier", float, 0.65), ("stop_head_lr_multiplier", float, 0.1),
    ("decoder_ffn_lr_multiplier", float, 0.30), ("decoder_attn_lr_multiplier", float, 0.15),
    ("variance_embedding_lr_multiplier", float, 0.15), ("qk_norm", bool, True),
    ("use_warmup", bool, True), ("warmup_steps", int, 1200), ("warmup_start_lr_ratio", float, 0.01),
    ("use_ema", bool, True), ("ema_decay", Optional[float], None), ("ema_half_life_epochs", float, 1.0),
    ("ema_update_every", int, 1),
    ("lr_T_0", int, 20), ("lr_T_mult", int, 2), ("lr_eta_min", float, 1e-6),
    ("n_mels", int, 80), ("hidden_dim", int, 512), ("n_encoder_layers", int, 6), ("n_decoder_layers", int, 6),
    ("n_heads", int, 8), ("encoder_ff_dim", int, 1536), ("decoder_ff_dim", int, 1536),
    ("encoder_dropout", float, 0.15), ("decoder_dropout", float, 0.20), ("decoder_input_dropout", float, 0.15),
    ("max_decoder_seq_len", int, 4000),
    ("use_stochastic_depth", bool, True), ("stochastic_depth_rate", float, 0.1), ("ffn_output_norm", bool, True),
    ("duration_loss_weight", float, 0.35), ("stop_token_loss_weight", float, 0.010),
    ("pitch_loss_weight", float, 1.0), ("energy_loss_weight", float, 1.0),
    ("pitch_huber_delta", float, 0.05), ("energy_huber_delta", float, 0.05),
    ("use_spec_augment", bool, True), ("spec_augment_time_mask_max", int, 5), ("spec_augment_freq_mask_max", int, 3),
    ("spec_augment_num_time_masks", int, 1), ("spec_augment_num_freq_masks", int, 2),
    ("spec_augment_start_epoch", int, 1),
    ("stop_token_pos_weight", float, 17.0), ("stop_token_smooth_tail", int, 6), ("stop_token_smooth_decay", float, 0.5),
    ("use_variance_predictor", bool, True), ("variance_filter_size", int, 256), ("variance_kernel_size", int, 3),
    ("variance_dropout", float, 0.1), ("n_variance_bins", int, 256),
    ("pitch_extract_fmin", float, 50.0), ("pitch_extract_fmax", float, 800.0),
    ("pitch_min", float, 0.0), ("pitch_max", float, 1.0), ("energy_min", float, 0.0), ("energy_max", float, 1.0),
    ("max_seq_length", int, 1800), ("sample_rate", int, 22050), ("hop_length", int, 256), ("win_length", int, 1024),
    ("n_fft", int, 1024), ("f_min", float, 0.0), ("f_max", float, 8000.0),
    ("use_speed_perturbation", bool, True), ("speed_perturb_range", float, 0.1), ("speed_perturb_prob", float, 0.5),
    ("num_workers", int, 0), ("pin_memory", bool, False),
    ("use_feature_cache", bool, True), ("feature_cache_dir", str, ""), ("precompute_features", bool, False),
    ("use_memory_cache", bool, True), ("enable_adaptive_memory", bool, True),
    ("use_dynamic_batching", bool, True), ("max_frames_per_batch", int, 15000), ("min_batch_size", int, 4),
    ("max_batch_size", int, 8),
    ("max_grad_norm", float, 1.5), ("projection_spike_clip_norm", float, 20.0),
    ("attention_spike_clip_norm", float, 4.0), ("ffn_spike_clip_norm", float, 3.0),
    ("encoder_ffn_spike_clip_norm", float, 8.0), ("stop_head_spike_clip_norm", float, 0.5),
    ("dec_ffn_max_weight_norm", float, 95.0), ("dec_ff0_linear1_max_weight_norm", float, 0.0),
    ("grad_explosion_warmup_steps", int, 400), ("grad_explosion_warmup_floor", float, 8000.0),
    ("grad_explosion_min_ema_steps", int, 100),
    ("save_every", int, 5), ("resume_checkpoint", str, "auto"),
    ("validation_split", float, 0.1), ("validation_interval", int, 1), ("early_stopping_patience", int, 15),
    ("early_stopping_min_delta", float, 0.001),
    ("use_mfa", bool, True), ("mfa_alignment_dir", str, "./mfa_output/alignments"),
    ("mfa_acoustic_model", str, "russian_mfa"), ("mfa_dictionary", str, "russian_mfa"),
    ("gradient_checkpointing", bool, True), ("checkpoint_segments", int, 2),
    ("auto_optimize_checkpointing", bool, False), ("target_memory_usage", float, 0.8),
    ("benchmark_checkpointing", bool, False),
    ("enable_profiling", bool, False), ("profile_epoch_start", int, 1), ("profile_wait_steps", int, 1),
    ("profile_warmup_steps", int, 1), ("profile_steps", int, 5), ("run_standalone_profiling", bool, False),
    ("verbose", bool, False), ("enable_interbatch_profiling", bool, False), ("interbatch_report_interval", int, 100),
    ("use_mixed_precision", bool, False),
    ("weight_decay", float, 0.04), ("ffn_weight_decay", float, 0.1), ("decoder_ffn_weight_decay", float, 0.35),
    ("adam_eps", float, 1e-8), ("adam_betas", tuple, (0.9, 0.999)), ("use_fused_adamw", Optional[bool], None),
    ("try_fused_adamw_on_mps", bool, True),
    ("use_torch_compile", bool, True), ("torch_compile_mode", str, "reduce-overhead"),
    ("torch_compile_dynamic", bool, True),
    # ---- new (MI355X engine) -------------------------------------------------------------------------------
    ("mixed_precision_dtype", str, "bfloat16"),   # MFMA arithmetic when use_mixed_precision is on
    ("dp_world_size", int, 1),                     # data-parallel ranks (one process per GPU, RCCL)
    ("replica_check_every", int, 500),             # data parallel: optimizer steps between parameter-checksum tripwires (0 = only at checkpoints)
]
N_REFERENCE_FIELDS = 133


def _post_init(self) -> None:
    if not self.feature_cache_dir:
        self.feature_cache_dir = str(Path(self.data_dir) / ".feature_cache")
    if self.checkpoint_segments < 1:
        self.checkpoint_segments = 1


TrainingConfig = dataclasses.make_dataclass(
    "TrainingConfig", [(n, t, dataclasses.field(default=d)) for n, t, d in _FIELDS],
    namespace={"__post_init__": _post_init, "__doc__": "Training configuration (reference-compatible fields)."})
TrainingConfig.__module__ = __name__
TrainingConfig.__qualname__ = "TrainingConfig"
