"""CPU oracle for the Kokoro acoustic-model train step.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (fp32, CPU) restatement of the
reference's algorithm for the hot path named in BASELINE.json; it is the *checker* for
the HIP engine, never the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference from
``/root/reference/src`` (build container only) and asserts that every function below
reproduces the reference's outputs, the 6 losses, all 308 gradients, the 10 optimizer
param groups and one full optimizer step; the resulting tensors are committed as
``tests/golden/*.npz`` and re-checked by ``tests/test_oracle_golden.py`` everywhere.

Each function cites the reference file:line it restates (paths relative to
``/root/reference/src/kokoro``).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# Dimensions (training/config.py:95-117,193-196)
# --------------------------------------------------------------------------------------
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


@dataclass
class StepHyper:
    """Step-driver knobs (training/config.py; defaults are the dataclass defaults)."""
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
    ema_decay: float = 0.9999
    ema_update_every: int = 1
    use_onecycle_lr: bool = True
    lr_T_0: int = 20
    lr_T_mult: int = 2
    lr_eta_min: float = 1e-6
    grad_explosion_ema_alpha: float = 0.95
    grad_explosion_abs_floor: float = 1000.0
    grad_explosion_multiplier: float = 3.0
    grad_explosion_warmup_steps: int = 400
    grad_explosion_warmup_floor: float = 8000.0
    grad_explosion_min_ema_steps: int = 100


# --------------------------------------------------------------------------------------
# Parameter / buffer tables in the reference's registration order
# (model/model.py:81-198, transformers.py:131-148,90-94,461-462,518-520,612,
#  variance_predictor.py:42-61,167-185)
# --------------------------------------------------------------------------------------
def _attn_names(prefix: str, H: int, dk: int) -> List[Tuple[str, Tuple[int, ...]]]:
    return [
        (f"{prefix}.w_q.weight", (H, H)),
        (f"{prefix}.w_k.weight", (H, H)),
        (f"{prefix}.w_v.weight", (H, H)),
        (f"{prefix}.w_o.weight", (H, H)),
        (f"{prefix}.w_o.bias", (H,)),
        (f"{prefix}.q_norm.weight", (dk,)),
        (f"{prefix}.k_norm.weight", (dk,)),
        (f"{prefix}.v_norm.weight", (dk,)),
    ]


def _ff_names(prefix: str, H: int, Fd: int) -> List[Tuple[str, Tuple[int, ...]]]:
    return [
        (f"{prefix}.linear1.weight", (2 * Fd, H)),
        (f"{prefix}.linear1.bias", (2 * Fd,)),
        (f"{prefix}.linear2.weight", (H, Fd)),
        (f"{prefix}.linear2.bias", (H,)),
        (f"{prefix}.output_norm.weight", (H,)),
    ]


def _varpred_names(prefix: str, H: int, Fv: int, k: int) -> List[Tuple[str, Tuple[int, ...]]]:
    return [
        (f"{prefix}.conv_layers.0.weight", (Fv, H, k)),
        (f"{prefix}.conv_layers.0.bias", (Fv,)),
        (f"{prefix}.conv_layers.1.weight", (Fv, Fv, k)),
        (f"{prefix}.conv_layers.1.bias", (Fv,)),
        (f"{prefix}.norms.0.weight", (Fv,)),
        (f"{prefix}.norms.0.bias", (Fv,)),
        (f"{prefix}.norms.1.weight", (Fv,)),
        (f"{prefix}.norms.1.bias", (Fv,)),
        (f"{prefix}.linear.weight", (1, Fv)),
        (f"{prefix}.linear.bias", (1,)),
    ]


def param_shapes(d: ModelDims) -> "OrderedDict[str, Tuple[int, ...]]":
    """308 parameter names → shapes in ``named_parameters()`` order."""
    H, dk = d.hidden, d.head_dim
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("text_embedding.weight", (d.vocab, H)),
        ("stress_embedding.weight", (3, H)),
    ]
    for i in range(d.enc_layers):
        p = f"transformer_encoder_layers.{i}"
        out += _attn_names(f"{p}.self_attn", H, dk)
        out += _ff_names(f"{p}.ff", H, d.enc_ff)
        out += [(f"{p}.norm1.weight", (H,)), (f"{p}.norm1.bias", (H,)),
                (f"{p}.norm2.weight", (H,)), (f"{p}.norm2.bias", (H,))]
    out += [("encoder_norm.weight", (H,)), ("encoder_norm.bias", (H,))]
    va = "duration_adaptor.variance_adaptor"
    for nm in ("duration_predictor", "pitch_predictor", "energy_predictor"):
        out += _varpred_names(f"{va}.{nm}", H, d.var_filter, d.var_kernel)
    out += [(f"{va}.pitch_embedding.weight", (d.var_bins, H)),
            (f"{va}.energy_embedding.weight", (d.var_bins, H))]
    out += [("mel_projection_in.weight", (H, d.mel)), ("mel_projection_in.bias", (H,))]
    for i in range(d.dec_layers):
        p = f"decoder.layers.{i}"
        out += _attn_names(f"{p}.self_attn", H, dk)
        out += _attn_names(f"{p}.cross_attn", H, dk)
        out += _ff_names(f"{p}.ff", H, d.dec_ff)
        for n in ("norm1", "norm2", "norm3"):
            out += [(f"{p}.{n}.weight", (H,)), (f"{p}.{n}.bias", (H,))]
    out += [("decoder.norm.weight", (H,)), ("decoder.norm.bias", (H,))]
    out += [("mel_projection_out.weight", (d.mel, H)), ("mel_projection_out.bias", (d.mel,))]
    out += [("stop_token_predictor.weight", (1, H)), ("stop_token_predictor.bias", (1,))]
    return OrderedDict(out)


def make_buffers(d: ModelDims) -> "OrderedDict[str, Tensor]":
    """Persistent buffers (positional_encoding.py:23-34; variance_predictor.py:181-182)."""
    position = torch.arange(d.max_len).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d.hidden, 2).float()
                         * (-torch.log(torch.tensor(10000.0)) / d.hidden))
    pe = torch.zeros(d.max_len, d.hidden)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    va = "duration_adaptor.variance_adaptor"
    return OrderedDict([
        ("positional_encoding.pe", pe.unsqueeze(0)),
        (f"{va}.pitch_bins", torch.linspace(0.0, 1.0, d.var_bins - 1)),
        (f"{va}.energy_bins", torch.linspace(0.0, 1.0, d.var_bins - 1)),
    ])


def state_dict_order(d: ModelDims) -> List[str]:
    """311 ``state_dict()`` keys in the reference's order (buffers sit with their module)."""
    names = list(param_shapes(d).keys())
    va = "duration_adaptor.variance_adaptor"
    out: List[str] = []
    for n in names:
        out.append(n)
        if n == "stress_embedding.weight":
            out.append("positional_encoding.pe")
    # pitch_bins/energy_bins are registered on VarianceAdaptor itself, so state_dict() lists
    # them before the adaptor's sub-modules' parameters.
    first_va = next(i for i, n in enumerate(out) if n.startswith(va + "."))
    out[first_va:first_va] = [f"{va}.pitch_bins", f"{va}.energy_bins"]
    return out


def rope_tables(seq_len: int, head_dim: int, base: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """cos/sin tables [seq, head_dim] (positional_encoding.py:129-150)."""
    theta = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    positions = torch.arange(seq_len, dtype=theta.dtype)
    freqs = torch.outer(positions, theta)
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def init_params(d: ModelDims, seed: int = 0) -> "OrderedDict[str, Tensor]":
    """Random init following the reference's formulas (model.py:85,93,174-198;
    transformers.py:97-103,176-183; variance_predictor.py:64-68,167-170).  Same
    distributions, not the same RNG stream — parity tests always load explicit weights."""
    g = torch.Generator().manual_seed(seed)
    P: "OrderedDict[str, Tensor]" = OrderedDict()

    def xavier(shape, gain=1.0):
        if len(shape) == 3:  # conv: fan_in = Cin*k, fan_out = Cout*k
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
            b = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * b
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


# --------------------------------------------------------------------------------------
# Length regulator — integer-exact (utils/lengths.py:16-96)
# --------------------------------------------------------------------------------------
def length_regulate_index(durations: np.ndarray, max_len: Optional[int] = None
                          ) -> Tuple[np.ndarray, np.ndarray, int]:
    """Return (idx[B,L] int64, lens[B] int64, L).

    ``idx[b,f]`` = #{j : cumsum(dur_b)[j] <= f} for f < lens[b]; -1 where f >= lens[b]
    (zero-filled frames).  L follows lengths.py:33-41,74-77: max_b Σdur clipped to
    ``max_len``; all-zero ⇒ max(1,max_len) or 1; result right-padded to ``max_len``.
    ``lens`` is Σdur_b clipped to L."""
    dur = np.asarray(durations)
    dur = np.trunc(dur).astype(np.int64) if dur.dtype.kind == "f" else dur.astype(np.int64)
    dur = np.clip(dur, 0, None)                       # lengths.py:31
    B, P = dur.shape
    cum = np.cumsum(dur, axis=1)
    total = cum[:, -1] if P > 0 else np.zeros(B, np.int64)
    max_expanded = int(total.max()) if B > 0 else 0
    if max_len is not None:
        max_expanded = min(max_expanded, int(max_len))
    if max_expanded == 0:
        L = max(1, int(max_len)) if max_len is not None else 1
        return np.full((B, L), -1, np.int64), np.zeros(B, np.int64), L
    L = max_expanded if max_len is None else max(max_expanded, int(max_len))
    f = np.arange(L, dtype=np.int64)
    idx = (cum[:, None, :] <= f[None, :, None]).sum(axis=2)          # (B, L)
    lens = np.minimum(total, max_expanded)
    idx = np.where(f[None, :] < lens[:, None], idx, -1)
    return idx.astype(np.int64), lens.astype(np.int64), L


def length_regulate(tokens: Tensor, durations: Tensor, max_len: Optional[int] = None) -> Tensor:
    """``vectorized_expand_tokens`` (lengths.py:16-96): gather + zero-fill, output detached."""
    idx, _, L = length_regulate_index(durations.detach().cpu().numpy(), max_len)
    idx_t = torch.from_numpy(idx).to(tokens.device)
    src = tokens.detach()
    safe = idx_t.clamp(min=0)
    if src.dim() == 3:
        out = torch.gather(src, 1, safe.unsqueeze(-1).expand(-1, -1, src.size(2)))
        out = out * (idx_t >= 0).unsqueeze(-1).to(out.dtype)
    else:
        out = torch.gather(src, 1, safe) * (idx_t >= 0).to(src.dtype)
    return out


# --------------------------------------------------------------------------------------
# Blocks
# --------------------------------------------------------------------------------------
def _rms_norm(x: Tensor, w: Tensor) -> Tensor:
    """nn.RMSNorm(eps=None): eps = finfo(dtype).eps (transformers.py:94,146-148)."""
    eps = torch.finfo(x.dtype).eps
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def _rotate_half(x: Tensor) -> Tensor:
    half = x.shape[-1] // 2
    return torch.cat((-x[..., half:], x[..., :half]), dim=-1)


def _drop(x: Tensor, p: float, on: bool) -> Tensor:
    return F.dropout(x, p, True) if (on and p > 0.0) else x


def attention(P: Dict[str, Tensor], prefix: str, xq: Tensor, xkv: Tensor, heads: int, *,
              rope: bool, causal: bool, key_mask: Optional[Tensor],
              p_drop: float = 0.0, drop_on: bool = False) -> Tensor:
    """MultiHeadAttentionImproved.forward, training path (transformers.py:211-437).

    q/k/v RMSNorm per head (:260-272), RoPE on q,k (:276-277), additive −inf masks
    (:299-316), softmax(QKᵀ/√d + bias)·V with dropout on the probabilities (:393-398),
    w_o with bias (:434)."""
    B, Sq, H = xq.shape
    Sk = xkv.shape[1]
    dk = H // heads
    q = F.linear(xq, P[f"{prefix}.w_q.weight"]).view(B, Sq, heads, dk).transpose(1, 2)
    k = F.linear(xkv, P[f"{prefix}.w_k.weight"]).view(B, Sk, heads, dk).transpose(1, 2)
    v = F.linear(xkv, P[f"{prefix}.w_v.weight"]).view(B, Sk, heads, dk).transpose(1, 2)
    v = _rms_norm(v, P[f"{prefix}.v_norm.weight"])
    q = _rms_norm(q, P[f"{prefix}.q_norm.weight"])
    k = _rms_norm(k, P[f"{prefix}.k_norm.weight"])
    if rope:
        cos, sin = (t.to(q.device) for t in rope_tables(max(Sq, Sk), dk))
        q = q * cos[:Sq] + _rotate_half(q) * sin[:Sq]
        k = k * cos[:Sk] + _rotate_half(k) * sin[:Sk]
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    if causal:
        scores = scores + torch.triu(torch.full((Sq, Sk), float("-inf"), device=scores.device), diagonal=1)
    if key_mask is not None:
        scores = scores.masked_fill(key_mask.bool()[:, None, None, :], float("-inf"))
    probs = _drop(torch.softmax(scores, dim=-1), p_drop, drop_on)
    ctx = torch.matmul(probs, v).transpose(1, 2).contiguous().view(B, Sq, H)
    return F.linear(ctx, P[f"{prefix}.w_o.weight"], P[f"{prefix}.w_o.bias"])


def glu_ffn(P: Dict[str, Tensor], prefix: str, x: Tensor, p_drop: float = 0.0,
            drop_on: bool = False) -> Tensor:
    """GLUFeedForward.forward (transformers.py:105-111): exact-erf GELU gate, RMSNorm out."""
    h = F.linear(x, P[f"{prefix}.linear1.weight"], P[f"{prefix}.linear1.bias"])
    gate, lin = h.chunk(2, dim=-1)
    y = F.linear(_drop(F.gelu(gate) * lin, p_drop, drop_on),
                 P[f"{prefix}.linear2.weight"], P[f"{prefix}.linear2.bias"])
    y = _rms_norm(y, P[f"{prefix}.output_norm.weight"])
    return _drop(y, p_drop, drop_on)


def _ln(P: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[f"{prefix}.weight"], P[f"{prefix}.bias"], 1e-5)


def _drop_path(x: Tensor, rate: float, on: bool) -> Tensor:
    """transformers.py:16-40."""
    if not on or rate == 0.0:
        return x
    keep = 1.0 - rate
    rnd = (keep + torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), device=x.device)).floor_()
    return x.div(keep) * rnd


def encoder_block(P, i: int, x: Tensor, key_mask: Tensor, heads: int, p_drop=0.0,
                  drop_on=False, dpr=0.0) -> Tensor:
    """ImprovedTransformerEncoderBlock.forward (transformers.py:468-489)."""
    p = f"transformer_encoder_layers.{i}"
    a = attention(P, f"{p}.self_attn", _ln(P, f"{p}.norm1", x), _ln(P, f"{p}.norm1", x), heads,
                  rope=True, causal=False, key_mask=key_mask, p_drop=p_drop, drop_on=drop_on)
    x = x + _drop(_drop_path(a, dpr, drop_on), p_drop, drop_on)
    f = glu_ffn(P, f"{p}.ff", _ln(P, f"{p}.norm2", x), p_drop, drop_on)
    return x + _drop(_drop_path(f, dpr, drop_on), p_drop, drop_on)


def decoder_block(P, i: int, x: Tensor, memory: Tensor, mem_mask: Tensor, heads: int,
                  p_drop=0.0, drop_on=False, dpr=0.0) -> Tensor:
    """ImprovedTransformerDecoderBlock.forward (transformers.py:543-583)."""
    p = f"decoder.layers.{i}"
    n1 = _ln(P, f"{p}.norm1", x)
    a = attention(P, f"{p}.self_attn", n1, n1, heads, rope=True, causal=True, key_mask=None,
                  p_drop=p_drop, drop_on=drop_on)
    x = x + _drop(_drop_path(a, dpr, drop_on), p_drop, drop_on)
    c = attention(P, f"{p}.cross_attn", _ln(P, f"{p}.norm2", x), memory, heads, rope=False,
                  causal=False, key_mask=mem_mask, p_drop=p_drop, drop_on=drop_on)
    x = x + _drop(_drop_path(c, dpr, drop_on), p_drop, drop_on)
    f = glu_ffn(P, f"{p}.ff", _ln(P, f"{p}.norm3", x), p_drop, drop_on)
    return x + _drop(_drop_path(f, dpr, drop_on), p_drop, drop_on)


def variance_predictor(P, prefix: str, x: Tensor, mask: Optional[Tensor], p_drop=0.0,
                       drop_on=False) -> Tensor:
    """VariancePredictor.forward/_forward_chunk (variance_predictor.py:70-115): independent
    512-frame chunks; GroupNorm(1,C) over (C×L_chunk) incl. padding; <2-frame chunk ⇒ zeros."""
    B, L, _ = x.shape
    outs = []
    for s in range(0, L, 512):
        xc = x[:, s:s + 512, :].transpose(1, 2)
        mc = mask[:, s:s + 512] if mask is not None else None
        if xc.size(2) < 2:
            o = torch.zeros(B, xc.size(2), dtype=x.dtype, device=x.device)
        else:
            h = xc
            for li in range(2):
                h = F.conv1d(h, P[f"{prefix}.conv_layers.{li}.weight"],
                             P[f"{prefix}.conv_layers.{li}.bias"], padding=1)
                h = F.group_norm(h, 1, P[f"{prefix}.norms.{li}.weight"],
                                 P[f"{prefix}.norms.{li}.bias"], 1e-5)
                h = _drop(F.relu(h), p_drop, drop_on)
            o = F.linear(h.transpose(1, 2), P[f"{prefix}.linear.weight"],
                         P[f"{prefix}.linear.bias"]).squeeze(-1)
        if mc is not None:
            o = o.masked_fill(mc, 0.0)
        outs.append(o)
    return torch.cat(outs, dim=1)


@dataclass
class DropCfg:
    """Dropout rates (training/config.py:108-121,195).  ``on=False`` is the parity mode."""
    on: bool = False
    encoder: float = 0.15
    decoder: float = 0.20
    decoder_input: float = 0.15
    variance: float = 0.10
    stochastic_depth: float = 0.1
    recompute: bool = False  # mimic the reference's activation checkpointing (cost only)


def forward(P: Dict[str, Tensor], Bf: Dict[str, Tensor], batch: Dict[str, Tensor],
            d: ModelDims, drop: Optional[DropCfg] = None,
            want: bool = False) -> Dict[str, Tensor]:
    """KokoroModel.forward_training (model/model.py:565-673) with the adaptor
    (variance_predictor.py:286-439) inlined.  Returns the 5 model outputs (+ intermediates
    when ``want``)."""
    drop = drop or DropCfg()
    on = drop.on
    H = d.hidden
    va = "duration_adaptor.variance_adaptor"
    ids = batch["phoneme_indices"]
    mel = batch["mel_specs"]
    B, T = mel.shape[0], mel.shape[1]
    Pn = ids.shape[1]
    inter: Dict[str, Tensor] = {}
    pe = Bf["positional_encoding.pe"][0]
    text_mask = ids == 0                                           # model.py:586-587

    def dpr(i, n):
        return (i / max(n - 1, 1)) * drop.stochastic_depth if on else 0.0

    # encode_text (model.py:375-388)
    x = F.embedding(ids, P["text_embedding.weight"]) * (H ** 0.5)
    if batch.get("stress_indices") is not None:
        x = x + F.embedding(batch["stress_indices"], P["stress_embedding.weight"], padding_idx=0)  # model.py:93
    x = _drop(x + pe[:Pn], drop.encoder, on)
    inter["enc_in"] = x
    for i in range(d.enc_layers):
        if drop.recompute and x.requires_grad:
            from torch.utils.checkpoint import checkpoint
            x = checkpoint(lambda xx, i=i: encoder_block(P, i, xx, text_mask, d.heads, drop.encoder,
                                                         on, dpr(i, d.enc_layers)), x, use_reentrant=False)
        else:
            x = encoder_block(P, i, x, text_mask, d.heads, drop.encoder, on, dpr(i, d.enc_layers))
        inter[f"enc_{i}"] = x
    enc = _ln(P, "encoder_norm", x)
    inter["enc_out"] = enc

    # VarianceAdaptor.forward (variance_predictor.py:338-439)
    dur_pred = variance_predictor(P, f"{va}.duration_predictor", enc, text_mask, drop.variance, on)
    dur = batch["phoneme_durations"].float()
    xf = length_regulate(enc, dur, None)                            # detached (lengths.py:30)
    if xf.size(1) < 3:
        xf = F.pad(xf, (0, 0, 0, 3 - xf.size(1)))
    lengths = dur.long().sum(dim=1)
    Lp = xf.size(1)
    frame_mask = torch.arange(Lp, device=lengths.device).unsqueeze(0) >= lengths.unsqueeze(1)
    pitch_pred = variance_predictor(P, f"{va}.pitch_predictor", xf, frame_mask, drop.variance, on)
    energy_pred = variance_predictor(P, f"{va}.energy_predictor", xf, frame_mask, drop.variance, on)

    def align(t):
        return t[:, :Lp] if t.size(1) >= Lp else F.pad(t, (0, Lp - t.size(1)))
    p_val, e_val = align(batch["pitches"]), align(batch["energies"])
    pb = torch.bucketize(p_val, Bf[f"{va}.pitch_bins"])
    eb = torch.bucketize(e_val, Bf[f"{va}.energy_bins"])
    adapted = xf + F.embedding(pb, P[f"{va}.pitch_embedding.weight"]) \
        + F.embedding(eb, P[f"{va}.energy_embedding.weight"])
    adapted = adapted.masked_fill(frame_mask.unsqueeze(-1), 0.0)
    inter["idx_pitch"], inter["idx_energy"] = pb, eb

    # align to mel length (model.py:607-628)
    if Lp > T:
        memory, mem_mask = adapted[:, :T], frame_mask[:, :T]
    elif Lp < T:
        memory = F.pad(adapted, (0, 0, 0, T - Lp))
        mem_mask = F.pad(frame_mask, (0, T - Lp), value=True)
    else:
        memory, mem_mask = adapted, frame_mask
    inter["memory"] = memory
    inter["mem_mask"] = mem_mask

    # decoder input (model.py:519-531)
    dec_in = F.pad(mel[:, :-1, :], (0, 0, 1, 0))
    y = F.linear(dec_in, P["mel_projection_in.weight"], P["mel_projection_in.bias"])
    y = _drop(y, drop.decoder_input, on)
    y = _drop(y + pe[:T], drop.encoder, on)
    inter["dec_in"] = y
    for i in range(d.dec_layers):
        if drop.recompute:
            from torch.utils.checkpoint import checkpoint
            y = checkpoint(lambda yy, i=i: decoder_block(P, i, yy, memory, mem_mask, d.heads, drop.decoder,
                                                         on, dpr(i, d.dec_layers)), y, use_reentrant=False)
        else:
            y = decoder_block(P, i, y, memory, mem_mask, d.heads, drop.decoder, on, dpr(i, d.dec_layers))
        inter[f"dec_{i}"] = y
    dec_out = _ln(P, "decoder.norm", y)
    inter["dec_out"] = dec_out

    # heads (model.py:561-562): stop head sees a detached input
    mel_pred = F.linear(dec_out, P["mel_projection_out.weight"], P["mel_projection_out.bias"])
    stop = F.linear(dec_out.detach(), P["stop_token_predictor.weight"],
                    P["stop_token_predictor.bias"]).squeeze(-1)
    out = {"mel": mel_pred, "log_dur": dur_pred, "stop": stop,
           "pitch": pitch_pred, "energy": energy_pred}
    if want:
        out.update({f"_{k}": v for k, v in inter.items()})
    return out


# --------------------------------------------------------------------------------------
# Inference: KokoroModel.forward_inference (model/model.py:676-790) + KokoroGenerator.generate
# (model/generator.py:24-127) with the decoder's incremental (KV-cache) path
# (transformers.py:237-277, 527-536, 543-583, 622-660)
# --------------------------------------------------------------------------------------
def encode_for_inference(P: Dict[str, Tensor], Bf: Dict[str, Tensor], ids: Tensor, stress: Optional[Tensor],
                         d: ModelDims) -> Dict[str, Tensor]:
    """encode_text + VarianceAdaptor.forward with no targets (variance_predictor.py:338-439): the model's own
    durations clamp(round(expm1(log_dur)), 0) drive the length regulator, its own clamped pitch / energy
    predictions pick the embeddings."""
    H = d.hidden
    va = "duration_adaptor.variance_adaptor"
    pe = Bf["positional_encoding.pe"][0]
    Pn = ids.shape[1]
    text_mask = ids == 0
    x = F.embedding(ids, P["text_embedding.weight"]) * (H ** 0.5)
    if stress is not None:
        x = x + F.embedding(stress, P["stress_embedding.weight"], padding_idx=0)
    x = x + pe[:Pn]
    for i in range(d.enc_layers):
        x = encoder_block(P, i, x, text_mask, d.heads)
    enc = _ln(P, "encoder_norm", x)
    log_dur = variance_predictor(P, f"{va}.duration_predictor", enc, text_mask)
    dur = torch.clamp(torch.round(torch.expm1(log_dur)), min=0)
    xf = length_regulate(enc, dur, None)
    if xf.size(1) < 3:
        xf = F.pad(xf, (0, 0, 0, 3 - xf.size(1)))
    lengths = dur.long().sum(dim=1)
    Lp = xf.size(1)
    frame_mask = torch.arange(Lp, device=lengths.device).unsqueeze(0) >= lengths.unsqueeze(1)
    pitch = variance_predictor(P, f"{va}.pitch_predictor", xf, frame_mask)
    energy = variance_predictor(P, f"{va}.energy_predictor", xf, frame_mask)
    pb = torch.bucketize(pitch.clamp(0.0, 1.0), Bf[f"{va}.pitch_bins"])
    eb = torch.bucketize(energy.clamp(0.0, 1.0), Bf[f"{va}.energy_bins"])
    memory = xf + F.embedding(pb, P[f"{va}.pitch_embedding.weight"]) + F.embedding(eb, P[f"{va}.energy_embedding.weight"])
    memory = memory.masked_fill(frame_mask.unsqueeze(-1), 0.0)
    return {"memory": memory, "mem_mask": frame_mask, "log_dur": log_dur, "durations": dur.long(), "pitch": pitch,
            "energy": energy, "enc": enc}


def _attn_step(P: Dict[str, Tensor], prefix: str, x: Tensor, heads: int, cache: Dict[str, Tensor]) -> Tensor:
    """Self-attention of ONE new position against the cache (transformers.py:237-253, 260-277).  The reference keeps
    raw K (normalised again every step — the same values) and normalised V; RoPE is applied with both offsets 0, so
    the keys are rotated by their absolute positions 0..t while the single query is rotated by position 0 (= not at
    all) — the inference-time behaviour of the reference, kept as is."""
    B, _, H = x.shape
    dk = H // heads
    q = F.linear(x, P[f"{prefix}.w_q.weight"]).view(B, 1, heads, dk).transpose(1, 2)
    k = F.linear(x, P[f"{prefix}.w_k.weight"]).view(B, 1, heads, dk).transpose(1, 2)
    v = F.linear(x, P[f"{prefix}.w_v.weight"]).view(B, 1, heads, dk).transpose(1, 2)
    v = _rms_norm(v, P[f"{prefix}.v_norm.weight"])
    cache["k_raw"] = k if "k_raw" not in cache else torch.cat([cache["k_raw"], k], dim=2)
    cache["v"] = v if "v" not in cache else torch.cat([cache["v"], v], dim=2)
    q = _rms_norm(q, P[f"{prefix}.q_norm.weight"])
    kk_ = _rms_norm(cache["k_raw"], P[f"{prefix}.k_norm.weight"])
    Sk = kk_.shape[2]
    cos, sin = rope_tables(Sk, dk)
    q = q * cos[:1] + _rotate_half(q) * sin[:1]
    kk_ = kk_ * cos[:Sk] + _rotate_half(kk_) * sin[:Sk]
    probs = torch.softmax(torch.matmul(q, kk_.transpose(-2, -1)) / math.sqrt(dk), dim=-1)
    ctx = torch.matmul(probs, cache["v"]).transpose(1, 2).contiguous().view(B, 1, H)
    return F.linear(ctx, P[f"{prefix}.w_o.weight"], P[f"{prefix}.w_o.bias"])


def generate(P: Dict[str, Tensor], Bf: Dict[str, Tensor], ids: Tensor, stress: Optional[Tensor], d: ModelDims, *,
             max_len: int = 4000, stop_threshold: float = 0.5, min_len_ratio: float = 0.7, min_len_floor: int = 12,
             max_len_ratio: float = 3.0, max_len_cap: int = 1600, post_expected_stop_threshold: float = 0.2,
             want: bool = False):
    """forward_inference: encode + expand by the predicted durations, then autoregressive decoding one frame at a
    time until the stop head fires (mean sigmoid over the batch > threshold, only from min_expected_length on; the
    threshold drops to min(threshold, post_expected) past the expected length), the last 30 frames are quiet
    (mean < -9.5), or max_expected_length frames exist.  The frame that triggers the stop is kept.  Output clamped
    to [-11.5, 2]."""
    enc = encode_for_inference(P, Bf, ids, stress, d)
    memory, mem_mask = enc["memory"], enc["mem_mask"]
    B, expected = memory.shape[0], memory.shape[1]
    min_expected = max(min_len_floor, int(expected * min_len_ratio))
    max_expected = min(max_len, max(expected + 80, int(expected * max_len_ratio)), max_len_cap)
    if max_expected <= min_expected:
        max_expected = min(max_len, min_expected + 1)
    pe = Bf["positional_encoding.pe"][0]
    caches = [dict() for _ in range(d.dec_layers)]
    frame = torch.zeros(B, 1, d.mel)
    frames, stop_probs = [], []
    for t in range(max_expected):
        y = F.linear(frame, P["mel_projection_in.weight"], P["mel_projection_in.bias"]) + pe[t:t + 1]
        for i in range(d.dec_layers):
            p = f"decoder.layers.{i}"
            y = y + _attn_step(P, f"{p}.self_attn", _ln(P, f"{p}.norm1", y), d.heads, caches[i])
            y = y + attention(P, f"{p}.cross_attn", _ln(P, f"{p}.norm2", y), memory, d.heads, rope=False, causal=False,
                              key_mask=mem_mask)
            y = y + glu_ffn(P, f"{p}.ff", _ln(P, f"{p}.norm3", y))
        out = _ln(P, "decoder.norm", y)
        frame = F.linear(out, P["mel_projection_out.weight"], P["mel_projection_out.bias"])
        stop = F.linear(out, P["stop_token_predictor.weight"], P["stop_token_predictor.bias"]).squeeze(-1)
        frames.append(frame)
        sp = float(torch.sigmoid(stop).mean())
        stop_probs.append(sp)
        if t >= min_expected:
            thr = stop_threshold if t < expected else min(stop_threshold, post_expected_stop_threshold)
            if sp > thr:
                break
            if len(frames) >= 30 and float(torch.cat(frames[-30:], dim=1).mean()) < -9.5:
                break
    mel = torch.cat(frames, dim=1).clamp(min=-11.5, max=2.0)
    if want:
        return mel, {**enc, "stop_probs": torch.tensor(stop_probs), "bounds": (min_expected, expected, max_expected)}
    return mel


# --------------------------------------------------------------------------------------
# Losses (training/losses.py:9-216; criteria trainer.py:410-444)
# --------------------------------------------------------------------------------------
def _huber(x: Tensor, y: Tensor, delta: float) -> Tensor:
    e = (x - y).abs()
    return torch.where(e <= delta, 0.5 * e * e, delta * (e - 0.5 * delta))


def _masked_mean(v: Tensor, m: Tensor) -> Tensor:
    m = m & torch.isfinite(v)
    return v[m].mean() if bool(m.any()) else torch.tensor(0.0, device=v.device)


def loss_sums(out: Dict[str, Tensor], batch: Dict[str, Tensor], hp: StepHyper) -> Tuple[List[Tensor], List[Tensor]]:
    """The five loss terms as (sum over valid elements, number of valid elements) in the order mel, dur, stop, pitch,
    energy — each reference term is such a masked mean (losses.py:60-199).  Data-parallel runs reduce these over the
    ranks and normalise by the global counts (kk_losses_finalize)."""
    mel_t = batch["mel_specs"]
    T, Pn = mel_t.size(1), batch["phoneme_durations"].size(1)
    mel_mask = torch.arange(T, device=mel_t.device).unsqueeze(0) < batch["mel_lengths"].unsqueeze(1)
    ph_mask = torch.arange(Pn, device=mel_t.device).unsqueeze(0) < batch["phoneme_lengths"].unsqueeze(1)
    l1 = (out["mel"] - mel_t).abs()
    m3 = mel_mask.unsqueeze(-1).expand_as(l1)
    tgt_dur = torch.log(batch["phoneme_durations"].float() + 1.0)
    ld = _huber(out["log_dur"], tgt_dur, hp.duration_huber_delta)
    dv = ph_mask & (batch["phoneme_durations"] > 0)
    z, y = out["stop"], batch["stop_token_targets"]
    # BCEWithLogits(pos_weight): -(pw*y*logσ(z) + (1-y)*logσ(-z))
    ls = -(hp.stop_token_pos_weight * y * F.logsigmoid(z) + (1 - y) * F.logsigmoid(-z))
    lp = _huber(out["pitch"][:, :T], batch["pitches"][:, :T], hp.pitch_huber_delta)
    le = _huber(out["energy"][:, :T], batch["energies"][:, :T], hp.energy_huber_delta)
    terms = [(l1, m3), (ld, dv), (ls, mel_mask), (lp, mel_mask), (le, mel_mask)]
    return [t[m].sum() for t, m in terms], [m.sum().to(torch.float64) for _, m in terms]


def losses_from_sums(sums: List[Tensor], counts: List[Tensor], hp: StepHyper) -> Tuple[Tensor, ...]:
    """(total, mel, dur, stop, pitch, energy) from (possibly globally reduced) sums and counts: clamps and weights of
    losses.py:200-216."""
    caps = (100.0, 100.0, 100.0, 10.0, 10.0)
    w = (1.0, hp.duration_loss_weight, hp.stop_token_loss_weight, hp.pitch_loss_weight, hp.energy_loss_weight)
    means = [(s / c.to(s.dtype) if float(c) > 0 else s * 0.0).clamp(max=cap) for s, c, cap in zip(sums, counts, caps)]
    total = sum(m * wk for m, wk in zip(means, w))
    return (total, *means)


def losses(out: Dict[str, Tensor], batch: Dict[str, Tensor], hp: StepHyper
           ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(total, mel, dur, stop, pitch, energy)."""
    mel_t = batch["mel_specs"]
    T, Pn = mel_t.size(1), batch["phoneme_durations"].size(1)
    mel_mask = torch.arange(T, device=mel_t.device).unsqueeze(0) < batch["mel_lengths"].unsqueeze(1)
    ph_mask = torch.arange(Pn, device=mel_t.device).unsqueeze(0) < batch["phoneme_lengths"].unsqueeze(1)

    l1 = (out["mel"] - mel_t).abs()
    loss_mel = _masked_mean(l1, mel_mask.unsqueeze(-1).expand_as(l1))

    tgt_dur = torch.log(batch["phoneme_durations"].float() + 1.0)
    ld = _huber(out["log_dur"], tgt_dur, hp.duration_huber_delta)
    dv = ph_mask & (batch["phoneme_durations"] > 0)
    loss_dur = ld[dv].mean() if bool(dv.any()) else torch.tensor(0.0, device=ld.device)

    z, y = out["stop"], batch["stop_token_targets"]
    # BCEWithLogits(pos_weight): -(pw*y*logσ(z) + (1-y)*logσ(-z))
    ls = -(hp.stop_token_pos_weight * y * F.logsigmoid(z) + (1 - y) * F.logsigmoid(-z))
    loss_stop = _masked_mean(ls, mel_mask)

    lp = _huber(out["pitch"][:, :T], batch["pitches"][:, :T], hp.pitch_huber_delta)
    loss_pitch = _masked_mean(lp, mel_mask)
    le = _huber(out["energy"][:, :T], batch["energies"][:, :T], hp.energy_huber_delta)
    loss_energy = _masked_mean(le, mel_mask)

    loss_mel = loss_mel.clamp(max=100.0)
    loss_dur = loss_dur.clamp(max=100.0)
    loss_stop = loss_stop.clamp(max=100.0)
    loss_pitch = loss_pitch.clamp(max=10.0)
    loss_energy = loss_energy.clamp(max=10.0)
    total = (loss_mel + loss_dur * hp.duration_loss_weight + loss_stop * hp.stop_token_loss_weight
             + loss_pitch * hp.pitch_loss_weight + loss_energy * hp.energy_loss_weight)
    return total, loss_mel, loss_dur, loss_stop, loss_pitch, loss_energy


def micro_batch_ok(out: Dict[str, Tensor], ls: Tuple[Tensor, ...]) -> bool:
    """The two guards of _execute_training_step: every element of the five predictions finite (trainer.py:3233-3256)
    and all six losses finite (:3274-3296).  A micro-batch failing either returns no step result: the trainer then drops
    the gradients accumulated so far, restarts the accumulation cycle and takes no optimizer step (:2304-2314)."""
    outs_ok = all(bool(torch.isfinite(out[k]).all()) for k in ("mel", "log_dur", "stop", "pitch", "energy"))
    return outs_ok and all(bool(torch.isfinite(x)) for x in ls)


# --------------------------------------------------------------------------------------
# Step driver: param groups, pre-clip, clip, AdamW, EMA, weight-norm, LR schedule
# --------------------------------------------------------------------------------------
GROUP_TYPES = ("encoder", "encoder", "decoder_other", "decoder_other", "decoder_attn",
               "decoder_attn", "decoder_ffn", "decoder_ffn", "variance_embed", "stop_head")


def param_group_of(name: str) -> int:
    """Index of the AdamW param group (0..9) a parameter lands in
    (training/trainer.py:503-642)."""
    enc_prefixes = ("text_embedding.", "stress_embedding.", "encoder_positional_encoding.",
                    "positional_encoding.", "transformer_encoder_layers.", "encoder_norm.")
    nd_sub = ("norm.weight", "norm.bias", "layer_norm.weight", "layer_norm.bias",
              "duration_adaptor.")
    no_decay = name.endswith(".bias") or any(s in name for s in nd_sub)
    if any(name.startswith(p) for p in enc_prefixes):
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
    """(lr multiplier, weight decay) of the 10 groups (trainer.py:591-642)."""
    e, a, f = hp.encoder_lr_multiplier, hp.decoder_attn_lr_multiplier, hp.decoder_ffn_lr_multiplier
    return [(e, 0.0), (e, hp.ffn_weight_decay), (1.0, 0.0), (1.0, hp.weight_decay),
            (a, hp.weight_decay), (a, 0.0), (f, hp.decoder_ffn_weight_decay), (f, 0.0),
            (hp.variance_embedding_lr_multiplier, 0.0), (hp.stop_head_lr_multiplier, 0.0)]


def preclip_max_norm(name: str, hp: StepHyper) -> Optional[float]:
    """Per-parameter spike clip ceiling or None (trainer.py:1340-1392)."""
    attn_frag = tuple(f".{a}.{w}.weight" for a in ("self_attn", "cross_attn")
                      for w in ("w_q", "w_k", "w_v", "w_o"))
    ffn_frag = (".linear1.weight", ".linear2.weight", ".linear1.bias", ".linear2.bias")
    if name in ("mel_projection_in.weight", "mel_projection_in.bias",
                "mel_projection_out.weight", "mel_projection_out.bias") and hp.projection_spike_clip_norm > 0:
        return hp.projection_spike_clip_norm
    if name in ("stop_token_predictor.weight", "stop_token_predictor.bias") and hp.stop_head_spike_clip_norm > 0:
        return hp.stop_head_spike_clip_norm
    if hp.attention_spike_clip_norm > 0 and (name.startswith("decoder.layers.")
                                             or name.startswith("transformer_encoder_layers.")) \
            and any(fr in name for fr in attn_frag):
        return hp.attention_spike_clip_norm
    if hp.encoder_ffn_spike_clip_norm > 0 and name.startswith("transformer_encoder_layers.") \
            and any(fr in name for fr in ffn_frag):
        return hp.encoder_ffn_spike_clip_norm
    if hp.ffn_spike_clip_norm > 0 and any(fr in name for fr in ffn_frag):
        return hp.ffn_spike_clip_norm
    return None


def is_weight_norm_target(name: str) -> bool:
    """The 24 FFN matrices projected by _apply_weight_norm_constraints (trainer.py:846-912)."""
    return ((name.startswith("decoder.layers.") or name.startswith("transformer_encoder_layers."))
            and (name.endswith(".ff.linear1.weight") or name.endswith(".ff.linear2.weight")))


def adaptive_loss_scale_and_clip(mel_length: int, max_duration: float, max_grad_norm: float
                                 ) -> Tuple[float, float]:
    """Batch-shape heuristics (trainer.py:2218-2242)."""
    scale, clip = 1.0, max_grad_norm
    soft = max(mel_length / 1400, max_duration / 150)
    if soft > 1.0:
        scale = min(scale, max(0.5, 1.0 / (soft ** 0.65)))
        clip = min(clip, max(0.3, 0.8 / (soft ** 0.35)))
    risk = max(mel_length / 1400, max_duration / 150)
    if risk > 1.0:
        scale = max(0.25, 1.0 / risk)
        clip = max(0.05, 0.5 / (risk ** 0.5))
    return scale, clip


def cosine_restart_factor(epoch: int, T_0: int, T_mult: int) -> float:
    """The "legacy" schedule (use_onecycle_lr = False, trainer.py:789-799): torch CosineAnnealingWarmRestarts(T_0, T_mult, eta_min)
    stepped ONCE PER EPOCH (trainer.py:2885-2887), no warm-up.  After `epoch` scheduler steps every param group runs at
    eta_min + (its initial lr - eta_min) * factor, factor = (1 + cos(pi * T_cur / T_i)) / 2 with T_cur / T_i from the restarts."""
    t_cur, t_i = int(epoch), int(T_0)
    while t_cur >= t_i:
        t_cur -= t_i
        t_i *= int(T_mult)
    return (1.0 + math.cos(math.pi * t_cur / t_i)) / 2.0


def legacy_group_lr(hp: "StepHyper", mult: float, epoch: int) -> float:
    return hp.lr_eta_min + (hp.learning_rate * mult - hp.lr_eta_min) * cosine_restart_factor(epoch, hp.lr_T_0, hp.lr_T_mult)


class LRSchedule:
    """Warmup + OneCycleLR exactly as the reference drives them
    (trainer.py:691-772, 1519-1575; torch OneCycleLR cos / three_phase=False).

    ``factor(k)`` is the base LR (before per-group multipliers) used by optimizer step k
    (0-based).  Quirk kept: step 0 runs at the full OneCycle initial LR because the manual
    warmup only takes effect after the first ``optimizer.step()``."""

    def __init__(self, hp: StepHyper, total_steps: int):
        self.hp = hp
        self.max_lr = hp.learning_rate * hp.max_lr_multiplier
        self.warmup_start = hp.learning_rate * hp.warmup_start_lr_ratio
        self.warmup_target = min(hp.learning_rate, self.max_lr)
        w = hp.warmup_steps if hp.use_warmup else 0
        if hp.use_warmup and w >= total_steps:                     # _apply_warmup_guard
            w = max(0, total_steps - 1)
        self.warmup_steps = w
        self.onecycle_steps = max(1, total_steps - w) if hp.use_warmup else total_steps
        self.div = max(1.0, float(hp.max_lr_multiplier)) if hp.use_warmup else 25.0
        self.final_div = 10000.0

    def _onecycle(self, step_num: int) -> float:
        total = self.onecycle_steps
        initial = self.max_lr / self.div
        min_lr = initial / self.final_div
        end1 = float(self.hp.pct_start * total) - 1
        end2 = total - 1

        def cos(start, end, pct):
            return (start - end) / 2.0 * (math.cos(math.pi * pct) + 1) + end
        if step_num <= end1:
            return cos(initial, self.max_lr, step_num / end1) if end1 > 0 else self.max_lr
        return cos(self.max_lr, min_lr, (step_num - end1) / (end2 - end1)) if end2 > end1 else min_lr

    def base_lr(self, k: int) -> float:
        if k == 0:
            return self._onecycle(0)
        j = k - 1                     # index of the scheduler call that set this LR
        if self.hp.use_warmup and j < self.warmup_steps:
            return self.warmup_start + (self.warmup_target - self.warmup_start) * (j / self.warmup_steps)
        s = j - self.warmup_steps + 1 if self.hp.use_warmup else j + 1
        return self._onecycle(min(s, self.onecycle_steps))


@dataclass
class OptState:
    step: int = 0
    m: Dict[str, Tensor] = field(default_factory=dict)
    v: Dict[str, Tensor] = field(default_factory=dict)


class ExplosionTracker:
    """Gradient-explosion tracker of the step boundary (trainer.py:914-925 set-up, :1315-1330 threshold, :2367-2405 use).

    threshold = floor(t) until the EMA of the total gradient norm has seen `min_ema_steps` norms, then
    max(floor(t), multiplier * EMA); floor(t) decays linearly from warmup_floor to abs_floor over the first
    `warmup_steps` COMPLETED optimizer steps.  A norm above the threshold caps the step's clip norm at 0.3
    ("emergency clipping"); the EMA then takes the norm in either way.  Python semantics kept: `nan > thr` is False and
    `max(floor, nan)` returns floor, so a non-finite norm never counts as an explosion but poisons the EMA for good
    (the step itself is skipped by the non-finite guard, trainer.py:2407-2463)."""

    def __init__(self, hp: "StepHyper"):
        self.alpha = hp.grad_explosion_ema_alpha
        self.abs_floor = hp.grad_explosion_abs_floor
        self.multiplier = hp.grad_explosion_multiplier
        self.warmup_steps = hp.grad_explosion_warmup_steps
        self.warmup_floor = hp.grad_explosion_warmup_floor
        self.min_ema_steps = hp.grad_explosion_min_ema_steps
        self.ema: Optional[float] = None
        self.ema_steps = 0
        self.streak = 0

    def threshold(self, steps_completed: int) -> Tuple[float, float, bool]:
        w = max(0, self.warmup_steps)
        if w > 0 and steps_completed < w:
            floor = self.warmup_floor - (self.warmup_floor - self.abs_floor) * (steps_completed / float(w))
        else:
            floor = self.abs_floor
        ready = self.ema_steps >= self.min_ema_steps
        ema_thr = 0.0 if self.ema is None else self.ema * self.multiplier
        return (floor if not ready else max(floor, ema_thr)), floor, ready

    def observe(self, total_norm: float, steps_completed: int, clip_norm: float) -> Tuple[float, bool]:
        """(clip norm for this step, exploding?) and the EMA / streak update."""
        thr, _, _ = self.threshold(steps_completed)
        exploding = total_norm > thr
        if exploding:
            self.streak += 1
            clip_norm = min(clip_norm, 0.3)
        else:
            self.streak = 0
        self.ema = total_norm if self.ema is None else self.alpha * self.ema + (1 - self.alpha) * total_norm
        self.ema_steps += 1
        return clip_norm, exploding


def optimizer_step(P: Dict[str, Tensor], G: Dict[str, Tensor], st: OptState, hp: StepHyper,
                   base_lr: float, clip_norm: float, ema: Optional[Dict[str, Tensor]] = None,
                   buffers: Optional[Dict[str, Tensor]] = None,
                   tracker: Optional[ExplosionTracker] = None, lr_of=None) -> Dict[str, float]:
    """One optimizer-step boundary, in the reference's order (trainer.py:2346-2477):
    pre-clip → total norm → [explosion tracker: emergency clip] → [non-finite gradients: skip] → global clip
    (runtime_policies.py:76; clip_grad_norm_) → AdamW → EMA over state_dict floats → FFN weight-norm projection.
    Mutates P, G, st, ema (and `tracker`).  `clip_norm` is the batch-shape adaptive clip norm
    (adaptive_loss_scale_and_clip) the reference enters the boundary with."""
    info: Dict[str, float] = {}
    with torch.no_grad():
        for n, g in G.items():                                     # S2
            mx = preclip_max_norm(n, hp)
            if mx is None or not bool(torch.isfinite(g).all()):
                continue
            nr = float(g.norm(2))
            if nr > mx:
                g.mul_(mx / (nr + 1e-12))
        total = math.sqrt(sum(float(g.norm(2)) ** 2 for g in G.values()))   # S3
        info["grad_norm"] = total
        if tracker is not None:
            clip_norm, info["exploding"] = tracker.observe(total, st.step, clip_norm)
        info["clip_norm"] = clip_norm
        if not all(bool(torch.isfinite(g).all()) for g in G.values()):      # trainer.py:2407-2463: skip, nothing moves
            info["skipped"] = True
            return info
        norms = torch.stack([g.norm(2) for g in G.values()])       # S4: clip_grad_norm_
        tn = torch.linalg.vector_norm(norms, 2)
        coef = torch.clamp(clip_norm / (tn + 1e-6), max=1.0)
        for g in G.values():
            g.mul_(coef)
        info["clip_coef"] = float(coef)
        st.step += 1                                               # S5: torch AdamW
        b1, b2 = hp.adam_betas
        bc1 = 1 - b1 ** st.step
        bc2s = math.sqrt(1 - b2 ** st.step)
        table = group_lr_mult_wd(hp)
        for n, p in P.items():
            mult, wd = table[param_group_of(n)]
            lr = base_lr * mult if lr_of is None else lr_of(mult)      # (lr_of: the legacy schedule, whose eta_min is not multiplied)
            g = G[n]
            if n not in st.m:
                st.m[n] = torch.zeros_like(p)
                st.v[n] = torch.zeros_like(p)
            p.mul_(1 - lr * wd)
            st.m[n].lerp_(g, 1 - b1)
            st.v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (st.v[n].sqrt() / bc2s).add_(hp.adam_eps)
            p.addcdiv_(st.m[n], denom, value=-(lr / bc1))
        # S7 (trainer.py:1491-1517): _update_ema runs after every SUCCESSFUL step and counts them in ema_updates; the weights move
        # when that count is a multiple of ema_update_every (0, N, 2N, ...).  st.step was incremented above: the count is st.step - 1.
        if ema is not None and (st.step - 1) % max(1, int(hp.ema_update_every)) == 0:
            dcy = hp.ema_decay
            for n, p in P.items():
                ema[n].mul_(dcy).add_(p, alpha=1 - dcy)
            for n, b in (buffers or {}).items():
                if n in ema:
                    ema[n].mul_(dcy).add_(b, alpha=1 - dcy)
        if hp.dec_ffn_max_weight_norm > 0:                         # S8
            for n, p in P.items():
                if is_weight_norm_target(n):
                    nr = float(p.norm(2))
                    if nr > hp.dec_ffn_max_weight_norm:
                        p.mul_(hp.dec_ffn_max_weight_norm / nr)
    return info


def grads_of(P: Dict[str, Tensor], Bf: Dict[str, Tensor], batch: Dict[str, Tensor], d: ModelDims,
             hp: StepHyper, loss_scale: float = 1.0, drop: Optional[DropCfg] = None
             ) -> Tuple[Dict[str, Tensor], Tuple[Tensor, ...], Dict[str, Tensor]]:
    """Forward + losses + backward by autograd on the restatement.  Returns
    (grads by name — zeros where autograd yields None, detached losses, outputs)."""
    Pg = OrderedDict((n, p.detach().clone().requires_grad_(True)) for n, p in P.items())
    out = forward(Pg, Bf, batch, d, drop)
    ls = losses(out, batch, hp)
    (ls[0] * loss_scale).backward()
    G = OrderedDict((n, (p.grad if p.grad is not None else torch.zeros_like(p)).detach())
                    for n, p in Pg.items())
    return G, tuple(x.detach() for x in ls), {k: v.detach() for k, v in out.items()}


# --------------------------------------------------------------------------------------
# Synthetic batches (dataset.py:32-64,581-606)
# --------------------------------------------------------------------------------------
def stop_targets(T: int, tail: int = 6, decay: float = 0.5) -> Tensor:
    """build_stop_token_targets (data/dataset.py:32-64)."""
    t = torch.zeros(T)
    if T > 0:
        n = min(tail + 1, T)
        t[T - n:T] = (decay ** torch.arange(n, dtype=torch.float32)).flip(0)
    return t


def synthetic_batch(B: int, T: int, Pn: int, d: ModelDims, seed: int = 1234,
                    ragged: bool = False) -> Dict[str, Tensor]:
    """Seeded synthetic padded batch with the collate_fn contract (dataset.py:871-921).
    Full-length samples unless ``ragged`` (then lengths ~ U[0.6,1]·max, zero padded)."""
    g = torch.Generator().manual_seed(seed)
    mel_len = torch.full((B,), T, dtype=torch.long)
    ph_len = torch.full((B,), Pn, dtype=torch.long)
    if ragged and B > 1:
        mel_len[1:] = (T * (0.6 + 0.4 * torch.rand(B - 1, generator=g))).long().clamp(min=4)
        ph_len[1:] = (Pn * (0.6 + 0.4 * torch.rand(B - 1, generator=g))).long().clamp(min=2)
    ids = torch.zeros(B, Pn, dtype=torch.long)
    stress = torch.zeros(B, Pn, dtype=torch.long)
    dur = torch.zeros(B, Pn, dtype=torch.long)
    mel = torch.zeros(B, T, d.mel)
    pitch = torch.zeros(B, T)
    energy = torch.zeros(B, T)
    stop = torch.zeros(B, T)
    for b in range(B):
        t, p = int(mel_len[b]), int(ph_len[b])
        ids[b, :p] = torch.randint(1, d.vocab, (p,), generator=g)
        stress[b, :p] = (torch.rand(p, generator=g) < 0.15).long()
        base = torch.full((p,), t // p, dtype=torch.long)
        base[: t % p] += 1                                          # dataset.py:581-606
        jit = torch.randint(-2, 3, (p,), generator=g)
        dd = (base + jit).clamp(min=1)
        diff = t - int(dd.sum())
        k = 0
        while diff != 0:                                            # re-normalise so Σ = t, min 1
            j = k % p
            if diff > 0:
                dd[j] += 1; diff -= 1
            elif dd[j] > 1:
                dd[j] -= 1; diff += 1
            k += 1
        dur[b, :p] = dd
        mel[b, :t] = (torch.randn(t, d.mel, generator=g) * 2 - 5).clamp(-11.5, 2.0)
        pv = torch.rand(t, generator=g)
        pv[torch.rand(t, generator=g) < 0.3] = 0.0
        pitch[b, :t] = pv
        energy[b, :t] = torch.rand(t, generator=g)
        stop[b, :t] = stop_targets(t)
    return {"mel_specs": mel, "phoneme_indices": ids, "stress_indices": stress,
            "phoneme_durations": dur, "stop_token_targets": stop, "pitches": pitch,
            "energies": energy, "mel_lengths": mel_len, "phoneme_lengths": ph_len}
