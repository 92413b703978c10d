"""ctypes binding of libkokoro_hip.so (declared in include/kokoro_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, a
RuntimeError carrying ``kk_last_error()`` is raised (same convention as the reference, which
raises from Python — SURVEY §8b "Error conventions").
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, List

import torch  # noqa: F401  -- must be imported BEFORE the .so is dlopen'ed: torch ships its own libamdhip64 and the
#                              HIP runtime our kernels launch through has to be the one torch initialised

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libkokoro_hip.so")

KK_MATH_F32, KK_MATH_BF16 = 0, 1
KK_SEG_ALIGN = 1024
OS = dict(SKIPPED=0, EXPL_EMA=1, EXPL_EMA_STEPS=2, EXPL_STREAK=3, LAST_GRAD_NORM=4, LAST_CLIP_COEF=5,
          LAST_SKIP=6, LAST_BASE_LR=7, LAST_CLIP_NORM=8, EXPL_EMA_VALID=9, ATTEMPT=10, BAD_SEG=11, BAD_COUNT=12,
          BAD_ATTEMPT=13, MICRO_BAD=14, MICRO_BAD_TOTAL=15, SIZE=16)


class KkLossCfg(C.Structure):
    _fields_ = [("w_dur", C.c_float), ("w_stop", C.c_float), ("w_pitch", C.c_float), ("w_energy", C.c_float),
                ("delta_dur", C.c_float), ("delta_pitch", C.c_float), ("delta_energy", C.c_float),
                ("pos_weight", C.c_float), ("loss_scale", C.c_float), ("adaptive", C.c_int)]


class KkOptCfg(C.Structure):
    _fields_ = [("learning_rate", C.c_double), ("max_lr", C.c_double), ("warmup_start_lr", C.c_double),
                ("warmup_target_lr", C.c_double), ("pct_start", C.c_double), ("div_factor", C.c_double),
                ("final_div_factor", C.c_double), ("warmup_steps", C.c_int64), ("onecycle_steps", C.c_int64),
                ("use_warmup", C.c_int), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("max_grad_norm", C.c_double), ("mel_length", C.c_int64), ("expl_alpha", C.c_double),
                ("expl_abs_floor", C.c_double), ("expl_multiplier", C.c_double), ("expl_warmup_floor", C.c_double),
                ("expl_warmup_steps", C.c_int64), ("expl_min_ema_steps", C.c_int64), ("ema_decay", C.c_double),
                ("max_weight_norm", C.c_double), ("ema_update_every", C.c_int64),
                ("legacy_schedule", C.c_int64), ("legacy_cos", C.c_double), ("eta_min", C.c_double)]


_P, _I, _L, _F, _D, _U = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_uint32

# name -> argtypes, in header order (tests/test_abi.py cross-checks arity against include/kokoro_hip.h)
class KkReduceDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst0", C.c_void_p), ("dst1", C.c_void_p), ("nblocks", C.c_int), ("ncols", C.c_int),
                ("split", C.c_int), ("stride", C.c_int)]


class KkAttnHeadNorm(C.Structure):
    _fields_ = [("raw", C.c_void_p), ("ldraw", C.c_int64), ("gain", C.c_void_p), ("partials", C.c_void_p),
                ("cos_t", C.c_void_p), ("sin_t", C.c_void_p), ("rope", C.c_int)]


def attn_headnorm(entries):
    """Host KkAttnHeadNorm array from [(raw, gain, partials, cos | None, sin | None)] (rope when cos is given)."""
    arr = (KkAttnHeadNorm * len(entries))()
    for d, (raw, gain, part, cos, sin) in zip(arr, entries):
        d.raw, d.ldraw, d.gain, d.partials = raw.data_ptr(), raw.stride(0), gain.data_ptr(), part.data_ptr()
        d.cos_t, d.sin_t, d.rope = (cos.data_ptr(), sin.data_ptr(), 1) if cos is not None else (0, 0, 0)
    return arr


class KkWgradDesc(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("lddy", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64), ("dw", C.c_void_p),
                ("lddw", C.c_int64), ("M", C.c_int64), ("N", C.c_int64), ("T", C.c_int64)]


def wgrad_table(entries):
    """Host KkWgradDesc array from [(dy [T, M] bf16, x [T, N] bf16, dW [M, N] fp32)] for kk_gemm_wgrad_group."""
    arr = (KkWgradDesc * len(entries))()
    for d, (dy, x, dw) in zip(arr, entries):
        d.dy, d.lddy, d.x, d.ldx, d.dw, d.lddw = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0)
        d.M, d.N, d.T = dy.shape[1], x.shape[1], dy.shape[0]
    return arr


class KkKeepSite(C.Structure):
    _fields_ = [("keep", C.c_void_p), ("site", C.c_uint32), ("p", C.c_float), ("B", C.c_int), ("heads", C.c_int), ("Sq", C.c_int),
                ("Sk", C.c_int), ("causal", C.c_int)]


def keep_sites(entries):
    """Host KkKeepSite array from [(keep buffer, site, p, B, heads, Sq, Sk, causal)] for kk_attn_keep_gen."""
    arr = (KkKeepSite * len(entries))()
    for d, (keep, site, p, B, heads, Sq, Sk, causal) in zip(arr, entries):
        d.keep, d.site, d.p, d.B, d.heads, d.Sq, d.Sk, d.causal = keep.data_ptr(), int(site), float(p), B, heads, Sq, Sk, 1 if causal else 0
    return arr


KK_ENC_MAX_LAYERS = 8


class KkEncLayer(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in (
        "w_qkv", "g_q", "g_k", "g_v", "w_o", "b_o", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2", "ffn_gain", "next_g", "next_b", "y1",
        "qkv_raw", "qkv_n", "ctx", "lse", "proj", "x_in", "xm", "y2", "mean2", "rstd2", "h1", "g", "f2", "rstd_f", "xo", "next_y",
        "next_mean", "next_rstd")] + [("next_y_bf16", C.c_int), ("site", C.c_uint32), ("p", C.c_float), ("dpr", C.c_float)])


class KkEncStack(C.Structure):
    _fields_ = [("B", C.c_int), ("S", C.c_int), ("H", C.c_int), ("F", C.c_int), ("heads", C.c_int), ("layers", C.c_int),
                ("key_mask", C.c_void_p), ("cos_t", C.c_void_p), ("sin_t", C.c_void_p), ("seed", C.c_void_p), ("sync", C.c_void_p),
                ("placement", C.c_int), ("trace", C.c_void_p), ("trace_wg", C.c_int),
                ("layer", KkEncLayer * KK_ENC_MAX_LAYERS)]


def enc_stack(B, S, H, F, heads, key_mask, cos, sin, seed, sync, layers, placement=0, trace=None, trace_wg=0) -> KkEncStack:
    """Host KkEncStack from per-layer dicts {field: tensor | int | float} (see include/kokoro_hip.h)."""
    d = KkEncStack()
    d.B, d.S, d.H, d.F, d.heads, d.layers, d.placement = B, S, H, F, heads, len(layers), placement
    d.key_mask = key_mask.data_ptr() if key_mask is not None else None
    d.cos_t, d.sin_t, d.sync = cos.data_ptr(), sin.data_ptr(), sync.data_ptr()
    d.seed = seed.data_ptr() if seed is not None else None
    d.trace, d.trace_wg = (trace.data_ptr() if trace is not None else None), trace_wg
    for L, src in zip(d.layer, layers):
        for name, v in src.items():
            setattr(L, name, v.data_ptr() if hasattr(v, "data_ptr") else v)
    return d


def pointer_table(tensors):
    """Host array of device pointers (for the C entry points that take `const float *const *`)."""
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def copy_many(pairs) -> None:
    """dst.copy_(src) for up to 16 (dst, src) pairs of contiguous same-size device tensors per launch."""
    for i in range(0, len(pairs), 16):
        part = pairs[i:i + 16]
        n = len(part)
        src = (C.c_void_p * n)(*[s.data_ptr() for _, s in part])
        dst = (C.c_void_p * n)(*[d.data_ptr() for d, _ in part])
        nbytes = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in part])
        call("kk_copy_many", src, dst, nbytes, n)


def zero_table(views):
    """(dst pointers, byte counts, n) host arrays for kk_zero_many from a list of contiguous device tensors (<= 160)."""
    n = len(views)
    return ((C.c_void_p * n)(*[v.data_ptr() for v in views]), (C.c_int64 * n)(*[v.numel() * v.element_size() for v in views]), n)


def reduce_table(entries, device) -> "torch.Tensor":
    """Device copy of a KkReduceDesc array from [(src, dst0, dst1 | None, nblocks, ncols, split[, row stride])]."""
    arr = (KkReduceDesc * len(entries))()
    for d, (src, dst0, dst1, nb, nc, split, *rest) in zip(arr, entries):
        d.src, d.dst0, d.dst1 = src.data_ptr(), dst0.data_ptr(), (dst1.data_ptr() if dst1 is not None else 0)
        d.nblocks, d.ncols, d.split, d.stride = nb, nc, split, (rest[0] if rest else 0)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


SIGNATURES: Dict[str, List[Any]] = {
    "kk_gemm": [_I, _I, _L, _L, _L, _F, _P, _L, _P, _L, _F, _P, _L, _P, _P, _L, _L, _I, _I, _I, _P],
    "kk_gemm_qkv_headnorm": [_L, _I, _I, _L, _P, _L, _P, _P, _P, _L, _P, _L, _I, _P, _I, _P, _P, _P],
    "kk_gemm_linear_glu": [_L, _L, _L, _P, _L, _P, _P, _P, _P, _L, _P, _U, _F, _P],
    "kk_gemm_dgrad_glu": [_L, _L, _L, _P, _L, _P, _P, _P, _P, _P, _U, _F, _P],
    "kk_gemm_dgrad_glu_blocks": [_L],
    "kk_gemm_wgrad_group": [_P, _I, _I, _I, _P, _P, _P, _P],
    "kk_colsum_acc": [_P, _L, _L, _L, _P, _I, _P],
    "kk_attn_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _P, _I, _F, _P, _U, _F, _I, _I, _P],
    "kk_attn_fwd_kb": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _P, _I, _F, _P, _U, _F, _I, _I, _P, _P],
    "kk_attn_fwd_rb": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _P, _I, _F, _P, _U, _F, _I, _I, _P, _P],
    "kk_attn_keep_gen": [_P, _I, _P, _I, _I, _P],
    "kk_attn_delta": [_P, _P, _P, _I, _I, _I, _L, _L, _I, _P],
    "kk_attn_bwd_dq": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _P, _I, _F, _P, _U, _F, _I, _I, _P, _L, _P, _P],
    "kk_zero_many": [_P, _P, _I, _P],
    "kk_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _P, _I, _F, _P, _U, _F, _I, _I, _P, _P, _P],
    "kk_attn_bwd_kb": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _P, _I, _F, _P, _U, _F, _I, _I, _P, _P, _P, _P],
    "kk_attn_bwd_ws": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _L, _P, _I, _F, _P, _U, _F, _I, _I, _P, _P, _P, _L, _P],
    "kk_attn_bwd_two_pass": [_I, _I, _I, _I, _I],
    "kk_gemm_dgrad_delta": [_L, _L, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _I, _I, _P],
    "kk_gemm_dgrad_delta_supported": [_L, _L, _L],
    "kk_attn_bwd_dkv": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _L, _L, _L, _P, _I, _F, _P, _U, _F, _I, _I, _P, _P],
    "kk_attn_bwd_blocks": [_I, _I, _I],
    "kk_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "kk_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _I, _I, _P],
    "kk_norm_bwd_blocks": [_L, _I],
    "kk_partials_reduce": [_P, _I, _I, _P],
    "kk_rmsnorm_fwd": [_P, _P, _P, _P, _P, _L, _I, _I, _P],
    "kk_rmsnorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "kk_headnorm_rope_fwd": [_P, _L, _P, _L, _L, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _P],
    "kk_headnorm_rope_bwd": [_P, _L, _P, _L, _P, _L, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P],
    "kk_headnorm_bwd_blocks": [_L, _I],
    "kk_glu_fwd": [_P, _P, _L, _I, _P, _U, _F, _I, _P],
    "kk_glu_bwd": [_P, _P, _P, _L, _I, _P, _U, _F, _I, _P],
    "kk_embed_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _U, _F, _P],
    "kk_embed_ln_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _U, _F, _P, _P, _P, _P, _I, _P, _P, _P],
    "kk_embed_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _U, _F, _P],
    "kk_length_regulate_index": [_P, _P, _P, _P, _I, _I, _I, _P],
    "kk_length_regulate_gather": [_P, _P, _P, _I, _I, _I, _I, _P],
    "kk_max_i64": [_P, _L, _P, _P],
    "kk_pad2d_f32": [_P, _L, _I, _P, _L, _I, _L, _P],
    "kk_frame_mask": [_P, _P, _I, _I, _P],
    "kk_im2col3_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "kk_im2col3_bwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "kk_groupnorm_relu_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _U, _F, _P],
    "kk_groupnorm_relu_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "kk_rowdot_fwd": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "kk_rowdot_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P],
    "kk_rowdot_bwd_blocks": [_L],
    "kk_bucket_embed_add_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "kk_regulate_embed_fwd": [_P] * 14 + [_I, _I, _I, _I, _I, _I, _P, _U, _I, _I, _I, _I, _P],
    "kk_bucket_embed_add_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "kk_bucket_sort_items": [_L, _I],
    "kk_bucket_sort": [_P, _P, _P, _L, _I, _P, _P, _P],
    "kk_bucket_embed_add_bwd_sorted": [_P, _P, _P, _P, _P, _L, _I, _I, _P],
    "kk_dropout_fwd": [_P, _P, _L, _P, _L, _I, _I, _P, _U, _F, _U, _F, _U, _F, _P],
    "kk_sublayer_out_fwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _P, _U, _F, _U, _F, _U, _F, _P],
    "kk_linear_tail_supported": [_L, _I, _I],
    "kk_linear_tail_pays": [_L, _I, _I],
    "kk_linear_tail_fwd": [_P, _L, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _P, _U, _F, _U, _F, _U, _F, _P],
    "kk_sublayer_in_bwd": [_P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _L, _I, _I, _P, _U, _F, _U, _F, _U, _F, _P],
    "kk_sublayer_in_bwd_blocks": [_L],
    "kk_encoder_stack_supported": [_I, _I, _I, _I, _I, _I],
    "kk_encoder_stack_workgroups": [],
    "kk_encoder_stack_fwd": [C.POINTER(KkEncStack), _P],
    "kk_dropout_bwd": [_P, _P, _L, _I, _I, _P, _U, _F, _U, _F, _U, _F, _I, _P],
    "kk_specaug": [_P, _I, _I, _I, _P, _U, _I, _I, _I, _I, _I, _P],
    "kk_ids_eq_zero": [_P, _P, _L, _P],
    "kk_shift_right": [_P, _P, _I, _I, _I, _P],
    "kk_decode_prologue": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "kk_decode_cache_append": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "kk_decode_epilogue": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "kk_losses_fwd": [_P] * 12 + [_I, _I, _I, _I, C.POINTER(KkLossCfg), _P, _P, _P, _P, _P, _I, _P],
    "kk_losses_finalize": [_P, C.POINTER(KkLossCfg), _P, _I, _P, _P, _P, _I, _P],
    "kk_losses_bwd": [_P] * 12 + [_I, _I, _I, _I, C.POINTER(KkLossCfg), _P, _P, _P, _P, _P, _P, _P],
    "kk_chain_begin": [_P],
    "kk_chain_launch": [_I, _P, _P, _I, _P],
    "kk_chain_abort": [_P],
    "kk_seg_sumsq": [_P, _P, _L, _P, _I, _P, _P, _I, _P],
    "kk_opt_prepare": [_P, _P, _P, _P, _I, _P, C.POINTER(KkOptCfg), _P, _P, _P, _P, _P, _P, _P, _P],
    "kk_adamw_ema": [_P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _F, _F, _F, _P, _I, _P, _I, _P],
    "kk_weight_norm_project": [_P, _P, _L, _P, _P, _P, _D, _P, _P],
    "kk_cast_f32_bf16": [_P, _P, _L, _P],
    "kk_comm_load": [C.c_char_p],
    "kk_comm_unique_id": [_P],
    "kk_comm_init": [_I, _I, _P],
    "kk_comm_world": [],
    "kk_comm_destroy": [],
    "kk_comm_reduce_bucket": [_P, _L, _I, _P],
    "kk_comm_reduce_ranges": [_P, _P, _P, _I, _I, _P],
    "kk_comm_loss_sync": [_P, _I, _P, _P],
    "kk_comm_reduce_scatter": [_P, _P, _L, _I, _P],
    "kk_comm_all_gather": [_P, _P, _L, _I, _P],
    "kk_cast_bf16_f32": [_P, _P, _L, _F, _P],
    "kk_cast_ranges": [_P, _P, _P, _P, _I, _I, _F, _P],
    "kk_copy_many": [_P, _P, _P, _I, _P],
    "kk_axpby": [_F, _P, _F, _P, _L, _P],
    "kk_timestamp": [_P, _P],
    "kk_mfma_probe": [_P, _P, _P],
}

_lib = None
_prof = None   # when a list: call() brackets every launch with events on the launch stream (bench.py roofline leg)


def profile_start() -> None:
    global _prof
    _prof = []


def profile_stop():
    """Returns [(name, args, ms)] for every call since profile_start() (synchronises)."""
    global _prof
    rec, _prof = _prof or [], None
    torch.cuda.synchronize()
    return [(n, a, s.elapsed_time(e)) for n, a, s, e in rec]


ABI_VERSION = 2        # include/kokoro_hip.h: KK_ABI_VERSION


def use_library(flavour: str) -> None:
    """Tools only: select the library flavour before the first call — "tuning" = libkokoro_hip_tuning.so (python -m
    kokoro_ruslan_amd.build --tuning: A/B switches and timing probes readable from the environment), "product" = the default."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_library() must be called before the library is loaded")
    LIB_PATH = os.path.join(HERE, "libkokoro_hip.so" if flavour == "product" else f"libkokoro_hip_{flavour}.so")


def load() -> C.CDLL:
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m kokoro_ruslan_amd.build` "
                           "(the MI355X engine has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.kk_last_error.restype = C.c_char_p
    lib.kk_abi_version.restype = C.c_int
    lib.kk_last_kernel.restype = C.c_char_p
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.kk_attn_bwd_ws_bytes.argtypes = [_I, _I, _I, _I]
    lib.kk_attn_bwd_ws_bytes.restype = C.c_int64
    lib.kk_attn_keep_bytes.argtypes = [_I, _I, _I, _I]
    lib.kk_attn_keep_bytes.restype = C.c_int64
    lib.kk_attn_warm_next.argtypes = [_P, _L, _P, _L]
    lib.kk_attn_warm_next.restype = C.c_int
    lib.kk_seg_sumsq_ws_bytes.argtypes = [_L]
    lib.kk_seg_sumsq_ws_bytes.restype = C.c_int64
    lib.kk_seg_sumsq_rec_offset.argtypes = []
    lib.kk_seg_sumsq_rec_offset.restype = C.c_int64
    lib.kk_seg_sumsq_rec_capacity.argtypes = []
    lib.kk_seg_sumsq_rec_capacity.restype = C.c_int
    if lib.kk_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {lib.kk_abi_version()}, this package binds version {ABI_VERSION} "
                           "(stale library: run `python -m kokoro_ruslan_amd.build --force`)")
    _lib = lib
    return lib


def _conv(a):
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    if isinstance(a, C.Structure):
        return C.byref(a)
    return a


def _tuning_hook(name: str):
    fn = getattr(load(), name, None)
    if fn is None:
        raise RuntimeError(f"{name} is a tools-only tuning hook (include/kokoro_hip_tuning.h): rebuild with "
                           "`python -m kokoro_ruslan_amd.build --tuning`, or set KK_GEMM16_TUNE before the first call")
    return fn


def gemm_tune(tm_threshold: int = 512, xcd_swizzle: int = 1) -> None:
    _tuning_hook("kk_gemm_tune")(tm_threshold, xcd_swizzle)


def gemm_tune16(enable: int = 1, thr128: int = 0, thr12864: int = 0, split_target: int = 0) -> None:
    """Benchmark hook for the bf16 x bf16 GEMM core: on/off, tile thresholds, split-K workgroup target (0 = keep)."""
    _tuning_hook("kk_gemm_tune16")(enable, thr128, thr12864, split_target)


def gemm_tune_group(split: int = 0) -> None:
    _tuning_hook("kk_gemm_tune_group")(split)


def last_kernel() -> str:
    """The kernel variant the last launching call of this thread took (kk_last_kernel): what tests assert routes with."""
    return load().kk_last_kernel().decode()


launches = 0          # kk.call invocations so far (the graph-capture driver uses it to drop empty segments)


def call(name: str, *args) -> None:
    """Invoke ``name`` with torch tensors (→ device pointers), scalars and cfg structs; the current
    torch stream is appended as the trailing ``stream`` argument."""
    global launches
    lib = load()
    launches += 1
    stream = torch.cuda.current_stream().cuda_stream
    if _prof is not None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(lib, name)(*[_conv(a) for a in args], stream)
        e.record()
        scalars = tuple(a for a in args if isinstance(a, (int, float)))
        for a in args:                                   # a grouped launch's problem sizes live in its descriptor table
            if isinstance(a, C.Array) and a._type_ is KkWgradDesc:
                scalars += tuple(v for d in a for v in (d.M, d.N, d.T))
        _prof.append((name, scalars, s, e))
    else:
        rc = getattr(lib, name)(*[_conv(a) for a in args], stream)
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {lib.kk_last_error().decode(errors='replace')}")
