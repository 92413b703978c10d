"""Static description of the Kokoro acoustic model and its step driver, as the engine needs it:
parameter names/shapes in the reference's registration order, persistent buffers, optimizer
param-group rules, per-parameter pre-clip classes and the hyper-parameters of the step.

Mirrors (file:line relative to /root/reference/src/kokoro):
  names/shapes      model/model.py:81-198, model/transformers.py:90-94,131-148,461-462,518-520,612,
                    model/variance_predictor.py:42-61,167-185
  param groups      training/trainer.py:503-642
  pre-clip classes  training/trainer.py:1340-1392
  weight-norm set   training/trainer.py:846-912
  defaults          training/config.py:16-352
This module is product code; the oracle keeps its own independent copy and tests compare the two
(and both against tests/golden/param_table.json, dumped from the reference itself).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, fields
from typing import Dict, List, Optional, Tuple

import torch

VA = "duration_adaptor.variance_adaptor"
GROUP_TYPES = ("encoder", "encoder", "decoder_other", "decoder_other", "decoder_attn", "decoder_attn",
               "decoder_ffn", "decoder_ffn", "variance_embed", "stop_head")


@dataclass
class ModelDims:
    vocab: int = 59
    mel: int = 80
    hidden: int = 512
    heads: int = 8
    enc_layers: int = 6
    dec_layers: int = 6
    enc_ff: int = 1536
    dec_ff: int = 1536
    var_filter: int = 256
    var_kernel: int = 3
    var_bins: int = 256
    max_len: int = 4000

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def validate(self) -> None:
        if self.head_dim != 64:
            raise ValueError(f"the gfx950 attention kernels are built for head_dim 64, got {self.head_dim}")
        if self.var_kernel != 3:
            raise ValueError("variance predictor kernel size must be 3")
        if 256 % self.var_filter or self.var_filter % 4:
            raise ValueError("variance_filter_size must divide 256 and be a multiple of 4")
        for v in (self.mel, self.hidden, self.enc_ff, self.dec_ff):
            if v % 4:
                raise ValueError("mel/hidden/ff dims must be multiples of 4 (16-byte vector loads)")


@dataclass
class StepHyper:
    """Step-driver knobs; defaults are the reference's TrainingConfig defaults."""
    learning_rate: float = 5.0e-5
    max_lr_multiplier: float = 1.0
    pct_start: float = 0.20
    encoder_lr_multiplier: float = 0.65
    stop_head_lr_multiplier: float = 0.1
    decoder_ffn_lr_multiplier: float = 0.30
    decoder_attn_lr_multiplier: float = 0.15
    variance_embedding_lr_multiplier: float = 0.15
    use_warmup: bool = True
    warmup_steps: int = 1200
    warmup_start_lr_ratio: float = 0.01
    weight_decay: float = 0.04
    ffn_weight_decay: float = 0.1
    decoder_ffn_weight_decay: float = 0.35
    adam_eps: float = 1e-8
    adam_betas: Tuple[float, float] = (0.9, 0.999)
    max_grad_norm: float = 1.5
    projection_spike_clip_norm: float = 20.0
    attention_spike_clip_norm: float = 4.0
    ffn_spike_clip_norm: float = 3.0
    encoder_ffn_spike_clip_norm: float = 8.0
    stop_head_spike_clip_norm: float = 0.5
    dec_ffn_max_weight_norm: float = 95.0
    duration_loss_weight: float = 0.35
    stop_token_loss_weight: float = 0.010
    pitch_loss_weight: float = 1.0
    energy_loss_weight: float = 1.0
    duration_huber_delta: float = 1.0
    pitch_huber_delta: float = 0.05
    energy_huber_delta: float = 0.05
    stop_token_pos_weight: float = 17.0
    use_ema: bool = True
    ema_decay: float = 0.9999
    ema_update_every: int = 1                  # config.py:87, trainer.py:1499-1502 (counted in successful optimizer steps)
    use_onecycle_lr: bool = True               # False: CosineAnnealingWarmRestarts per EPOCH, no warm-up (trainer.py:789-799, 2885-2887)
    lr_T_0: int = 20
    lr_T_mult: int = 2
    lr_eta_min: float = 1e-6
    grad_explosion_ema_alpha: float = 0.95
    grad_explosion_abs_floor: float = 1000.0
    grad_explosion_multiplier: float = 3.0
    grad_explosion_warmup_steps: int = 400
    grad_explosion_warmup_floor: float = 8000.0
    grad_explosion_min_ema_steps: int = 100
    gradient_accumulation_steps: int = 2
    # regularisation (config.py:108-121,156-161,195); active only when the engine's `train_dropout` flag is on
    encoder_dropout: float = 0.15
    decoder_dropout: float = 0.20
    decoder_input_dropout: float = 0.15
    variance_dropout: float = 0.1
    use_stochastic_depth: bool = True
    stochastic_depth_rate: float = 0.1
    use_spec_augment: bool = True
    spec_augment_time_mask_max: int = 5
    spec_augment_freq_mask_max: int = 3
    spec_augment_num_time_masks: int = 1
    spec_augment_num_freq_masks: int = 2
    spec_augment_start_epoch: int = 1

    @classmethod
    def from_config(cls, cfg) -> "StepHyper":
        """Pick the fields a TrainingConfig-like object carries (getattr with our defaults)."""
        kw = {}
        for f in fields(cls):
            if hasattr(cfg, f.name) and getattr(cfg, f.name) is not None:
                kw[f.name] = getattr(cfg, f.name)
        return cls(**kw)


def _attn(prefix: str, H: int, dk: int):
    return [(f"{prefix}.w_q.weight", (H, H)), (f"{prefix}.w_k.weight", (H, H)), (f"{prefix}.w_v.weight", (H, H)),
            (f"{prefix}.w_o.weight", (H, H)), (f"{prefix}.w_o.bias", (H,)), (f"{prefix}.q_norm.weight", (dk,)),
            (f"{prefix}.k_norm.weight", (dk,)), (f"{prefix}.v_norm.weight", (dk,))]


def _ff(prefix: str, H: int, Fd: int):
    return [(f"{prefix}.linear1.weight", (2 * Fd, H)), (f"{prefix}.linear1.bias", (2 * Fd,)),
            (f"{prefix}.linear2.weight", (H, Fd)), (f"{prefix}.linear2.bias", (H,)),
            (f"{prefix}.output_norm.weight", (H,))]


def _varpred(prefix: str, H: int, Fv: int, k: int):
    return [(f"{prefix}.conv_layers.0.weight", (Fv, H, k)), (f"{prefix}.conv_layers.0.bias", (Fv,)),
            (f"{prefix}.conv_layers.1.weight", (Fv, Fv, k)), (f"{prefix}.conv_layers.1.bias", (Fv,)),
            (f"{prefix}.norms.0.weight", (Fv,)), (f"{prefix}.norms.0.bias", (Fv,)),
            (f"{prefix}.norms.1.weight", (Fv,)), (f"{prefix}.norms.1.bias", (Fv,)),
            (f"{prefix}.linear.weight", (1, Fv)), (f"{prefix}.linear.bias", (1,))]


def param_shapes(d: ModelDims) -> "OrderedDict[str, Tuple[int, ...]]":
    H, dk = d.hidden, d.head_dim
    out = [("text_embedding.weight", (d.vocab, H)), ("stress_embedding.weight", (3, H))]
    for i in range(d.enc_layers):
        p = f"transformer_encoder_layers.{i}"
        out += _attn(f"{p}.self_attn", H, dk) + _ff(f"{p}.ff", H, d.enc_ff)
        out += [(f"{p}.norm1.weight", (H,)), (f"{p}.norm1.bias", (H,)), (f"{p}.norm2.weight", (H,)), (f"{p}.norm2.bias", (H,))]
    out += [("encoder_norm.weight", (H,)), ("encoder_norm.bias", (H,))]
    for nm in ("duration_predictor", "pitch_predictor", "energy_predictor"):
        out += _varpred(f"{VA}.{nm}", H, d.var_filter, d.var_kernel)
    out += [(f"{VA}.pitch_embedding.weight", (d.var_bins, H)), (f"{VA}.energy_embedding.weight", (d.var_bins, H))]
    out += [("mel_projection_in.weight", (H, d.mel)), ("mel_projection_in.bias", (H,))]
    for i in range(d.dec_layers):
        p = f"decoder.layers.{i}"
        out += _attn(f"{p}.self_attn", H, dk) + _attn(f"{p}.cross_attn", H, dk) + _ff(f"{p}.ff", H, d.dec_ff)
        for n in ("norm1", "norm2", "norm3"):
            out += [(f"{p}.{n}.weight", (H,)), (f"{p}.{n}.bias", (H,))]
    out += [("decoder.norm.weight", (H,)), ("decoder.norm.bias", (H,))]
    out += [("mel_projection_out.weight", (d.mel, H)), ("mel_projection_out.bias", (d.mel,))]
    out += [("stop_token_predictor.weight", (1, H)), ("stop_token_predictor.bias", (1,))]
    return OrderedDict(out)


def buffer_shapes(d: ModelDims) -> "OrderedDict[str, Tuple[int, ...]]":
    return OrderedDict([("positional_encoding.pe", (1, d.max_len, d.hidden)),
                        (f"{VA}.pitch_bins", (d.var_bins - 1,)), (f"{VA}.energy_bins", (d.var_bins - 1,))])


def make_buffers(d: ModelDims) -> "OrderedDict[str, torch.Tensor]":
    """positional_encoding.py:23-34; variance_predictor.py:181-182 (same torch ops ⇒ same bits)."""
    position = torch.arange(d.max_len).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d.hidden, 2).float() * (-torch.log(torch.tensor(10000.0)) / d.hidden))
    pe = torch.zeros(d.max_len, d.hidden)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return OrderedDict([("positional_encoding.pe", pe.unsqueeze(0)),
                        (f"{VA}.pitch_bins", torch.linspace(0.0, 1.0, d.var_bins - 1)),
                        (f"{VA}.energy_bins", torch.linspace(0.0, 1.0, d.var_bins - 1))])


def rope_tables(seq_len: int, head_dim: int = 64, base: float = 10000.0):
    """positional_encoding.py:129-150."""
    theta = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    freqs = torch.outer(torch.arange(seq_len, dtype=theta.dtype), theta)
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos().contiguous(), emb.sin().contiguous()


SEG_ALIGN = 1024      # == KK_SEG_ALIGN (include/kokoro_hip.h): arena segments start on multiples of this many elements


def arena_layout(d: ModelDims):
    """(names, shapes, offsets, total) of the flat parameter arena: one 1024-aligned segment per tensor.  Physical order: the K
    and V projections of all decoder cross-attention layers first and contiguous (one GEMM projects the memory for every
    layer), everything else in state-dict order.  Pure host arithmetic (the data-parallel bucket plan needs it on CPU)."""
    segs = list(param_shapes(d).items()) + list(buffer_shapes(d).items())
    is_ckv = lambda n: n.endswith(".cross_attn.w_k.weight") or n.endswith(".cross_attn.w_v.weight")
    segs = [x for x in segs if is_ckv(x[0])] + [x for x in segs if not is_ckv(x[0])]
    offset, off = {}, 0
    for n, shp in segs:
        offset[n] = off
        off += -(-math.prod(shp) // SEG_ALIGN) * SEG_ALIGN
    return [n for n, _ in segs], dict(segs), offset, off


def state_dict_order(d: ModelDims) -> List[str]:
    out: List[str] = []
    for n in param_shapes(d):
        out.append(n)
        if n == "stress_embedding.weight":
            out.append("positional_encoding.pe")
    first = next(i for i, n in enumerate(out) if n.startswith(VA + "."))
    out[first:first] = [f"{VA}.pitch_bins", f"{VA}.energy_bins"]
    return out


def param_group_of(name: str) -> int:
    enc = ("text_embedding.", "stress_embedding.", "encoder_positional_encoding.", "positional_encoding.",
           "transformer_encoder_layers.", "encoder_norm.")
    nd_sub = ("norm.weight", "norm.bias", "layer_norm.weight", "layer_norm.bias", "duration_adaptor.")
    no_decay = name.endswith(".bias") or any(s in name for s in nd_sub)
    if any(name.startswith(p) for p in enc):
        return 1 if (".ff." in name and not no_decay) else 0
    if name in ("stop_token_predictor.weight", "stop_token_predictor.bias"):
        return 9
    if no_decay:
        if "pitch_embedding." in name or "energy_embedding." in name:
            return 8
        if ".ff." in name:
            return 7
        if ".self_attn." in name or ".cross_attn." in name:
            return 5
        return 2
    if ".ff." in name or ".ff" in name:
        return 6
    if ".self_attn." in name or ".cross_attn." in name:
        return 4
    return 3


def group_lr_mult_wd(hp: StepHyper) -> List[Tuple[float, float]]:
    e, a, f = hp.encoder_lr_multiplier, hp.decoder_attn_lr_multiplier, hp.decoder_ffn_lr_multiplier
    return [(e, 0.0), (e, hp.ffn_weight_decay), (1.0, 0.0), (1.0, hp.weight_decay), (a, hp.weight_decay), (a, 0.0),
            (f, hp.decoder_ffn_weight_decay), (f, 0.0), (hp.variance_embedding_lr_multiplier, 0.0),
            (hp.stop_head_lr_multiplier, 0.0)]


_ATTN_FRAG = tuple(f".{a}.{w}.weight" for a in ("self_attn", "cross_attn") for w in ("w_q", "w_k", "w_v", "w_o"))
_FFN_FRAG = (".linear1.weight", ".linear2.weight", ".linear1.bias", ".linear2.bias")


def preclip_max_norm(name: str, hp: StepHyper) -> Optional[float]:
    if name in ("mel_projection_in.weight", "mel_projection_in.bias", "mel_projection_out.weight",
                "mel_projection_out.bias") and hp.projection_spike_clip_norm > 0:
        return hp.projection_spike_clip_norm
    if name in ("stop_token_predictor.weight", "stop_token_predictor.bias") and hp.stop_head_spike_clip_norm > 0:
        return hp.stop_head_spike_clip_norm
    layered = name.startswith("decoder.layers.") or name.startswith("transformer_encoder_layers.")
    if hp.attention_spike_clip_norm > 0 and layered and any(f in name for f in _ATTN_FRAG):
        return hp.attention_spike_clip_norm
    if hp.encoder_ffn_spike_clip_norm > 0 and name.startswith("transformer_encoder_layers.") and any(f in name for f in _FFN_FRAG):
        return hp.encoder_ffn_spike_clip_norm
    if hp.ffn_spike_clip_norm > 0 and any(f in name for f in _FFN_FRAG):
        return hp.ffn_spike_clip_norm
    return None


def is_weight_norm_target(name: str) -> bool:
    return ((name.startswith("decoder.layers.") or name.startswith("transformer_encoder_layers."))
            and (name.endswith(".ff.linear1.weight") or name.endswith(".ff.linear2.weight")))


def init_params(d: ModelDims, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Random initialisation with the reference's distributions (model.py:85,93,174-198;
    transformers.py:97-103,176-183; variance_predictor.py:64-68,167-170)."""
    g = torch.Generator().manual_seed(seed)
    P: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def xavier(shape, gain=1.0):
        if len(shape) == 3:
            fan_out, fan_in = shape[0] * shape[2], shape[1] * shape[2]
        else:
            fan_out, fan_in = shape[0], shape[1]
        a = gain * math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=g) * 2 - 1) * a

    for name, shape in param_shapes(d).items():
        if name == "text_embedding.weight":
            t = torch.randn(shape, generator=g) / math.sqrt(d.hidden)
        elif name == "stress_embedding.weight":
            t = torch.randn(shape, generator=g)
            t[0].zero_()
        elif name.endswith("_embedding.weight"):
            t = torch.randn(shape, generator=g)
        elif name.endswith("duration_predictor.linear.bias"):
            t = torch.full(shape, math.log1p(5))
        elif ".conv_layers." in name and name.endswith(".bias"):
            fan_in = d.var_kernel * (d.hidden if ".conv_layers.0." in name else d.var_filter)
            t = (torch.rand(shape, generator=g) * 2 - 1) * (1.0 / math.sqrt(fan_in))
        elif name.endswith(".bias"):
            t = torch.zeros(shape)
        elif "norm" in name and name.endswith(".weight"):
            t = torch.ones(shape)
        elif name.endswith("ff.linear2.weight"):
            t = xavier(shape, 0.5)
        else:
            t = xavier(shape)
        P[name] = t.float()
    return P


def cosine_restart_position(epoch: int, T_0: int, T_mult: int) -> Tuple[int, int]:
    """(T_cur, T_i) of torch's CosineAnnealingWarmRestarts after `epoch` scheduler steps.  Raises ValueError for the values torch's
    constructor rejects (T_0 <= 0, T_mult < 1) instead of looping forever on them."""
    if int(T_0) != T_0 or int(T_0) <= 0:
        raise ValueError(f"Expected positive integer T_0, but got {T_0}")
    if int(T_mult) != T_mult or int(T_mult) < 1:
        raise ValueError(f"Expected integer T_mult >= 1, but got {T_mult}")
    t_cur, t_i = int(epoch), int(T_0)
    while t_cur >= t_i:
        t_cur -= t_i
        t_i *= int(T_mult)
    return t_cur, t_i


def cosine_restart_factor(epoch: int, T_0: int, T_mult: int) -> float:
    """(1 + cos(pi * T_cur / T_i)) / 2 of torch's CosineAnnealingWarmRestarts after `epoch` scheduler steps: the reference's legacy
    schedule (use_onecycle_lr = False) steps it once per epoch (trainer.py:789-799, 2885-2887)."""
    t_cur, t_i = cosine_restart_position(epoch, T_0, T_mult)
    return (1.0 + math.cos(math.pi * t_cur / t_i)) / 2.0


def lr_schedule_consts(hp: StepHyper, total_steps: int) -> Dict[str, float]:
    """Constants of the warmup + OneCycleLR pair (trainer.py:691-772; _apply_warmup_guard :1638-1652)."""
    max_lr = hp.learning_rate * hp.max_lr_multiplier
    w = hp.warmup_steps if hp.use_warmup else 0
    if hp.use_warmup and w >= total_steps:
        w = max(0, total_steps - 1)
    return dict(learning_rate=hp.learning_rate, max_lr=max_lr,
                warmup_start_lr=hp.learning_rate * hp.warmup_start_lr_ratio,
                warmup_target_lr=min(hp.learning_rate, max_lr), pct_start=hp.pct_start,
                div_factor=max(1.0, float(hp.max_lr_multiplier)) if hp.use_warmup else 25.0,
                final_div_factor=10000.0, warmup_steps=w,
                onecycle_steps=max(1, total_steps - w) if hp.use_warmup else max(1, total_steps),
                use_warmup=int(hp.use_warmup))
