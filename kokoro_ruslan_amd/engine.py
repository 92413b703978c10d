"""MI355X engine for the Kokoro acoustic-model train step.

Host side of the hot path: owns a flat fp32 parameter arena (params, grads, Adam m/v, EMA as contiguous
HBM slabs with a segment table), per-shape activation workspaces, and the explicit forward / backward /
optimizer kernel sequences.  Autograd is not used: the model is a fixed DAG, so the backward is a
hand-written reverse sequence of the same C-ABI kernels (kokoro_ruslan_amd.lib → libkokoro_hip.so).
Nothing in a step synchronises with the host, so a whole step can be captured in one hipGraph.

What it restates (reference file:line, relative to /root/reference/src/kokoro):
  forward        model/model.py:565-673 (forward_training) with encode_text :358-388, VarianceAdaptor.forward
                 model/variance_predictor.py:286-439, decoder model/transformers.py:543-583,622-662
  losses         training/losses.py:9-216
  step driver    training/trainer.py:2218-2242,2346-2477 + runtime_policies.py:14-87 (see csrc/kk_optim.hip)
Quirks kept on purpose: the length-regulated memory is detached (no gradient from the decoder into
the text encoder, facts 5); key padding = (phoneme id == 0) (fact 6); GroupNorm(1,C) statistics per 512-frame
chunk including padding (fact 7); stop head sees a detached decoder output (model.py:562).
Expanded length T' = max_b Σdur != mel length T (model/model.py:607-628): see forward_backward(expanded_len=...).
Activation memory: one workspace sized for the largest batch seen (see _buf), whatever the number of batch shapes.
"""
from __future__ import annotations

import contextlib
import math
import os
import threading
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

from . import lib as kk
from . import spec
from .spec import VA, ModelDims, StepHyper

CHUNK = 512   # variance_predictor.py:77
_ITEMSIZE = {torch.float32: 4, torch.bfloat16: 2, torch.float64: 8, torch.int64: 8, torch.int32: 4, torch.uint8: 1}


def _b16(t) -> int:
    """1 when the tensor is stored as bf16 (the C ABI's *_bf16 flags are derived from the tensors themselves)."""
    return 1 if t is not None and t.dtype == torch.bfloat16 else 0


class Arena:
    """Flat fp32 slabs p / g / m / v / ema with one 1024-aligned, zero-padded segment per tensor."""

    def __init__(self, dims: ModelDims, hp: StepHyper, device: torch.device, shadow_bf16: bool = False):
        self.dims, self.device = dims, device
        self.param_names = list(spec.param_shapes(dims).keys())
        # physical order (spec.arena_layout): the K and V projections of all decoder cross-attention layers first and
        # contiguous, so the memory is projected for every layer by ONE GEMM against a [layers*2H, H] matrix (and its two
        # gradients likewise); everything else in state-dict order.  Names, not offsets, are the interface.
        assert spec.SEG_ALIGN == kk.KK_SEG_ALIGN
        self.names, self.shapes, self.offset, off = spec.arena_layout(dims)
        segs = [(n, self.shapes[n]) for n in self.names]
        self.total = off
        self.nblocks = off // kk.KK_SEG_ALIGN
        self.nseg = len(segs)
        z = lambda: torch.zeros(off, dtype=torch.float32, device=device)
        self.p, self.g, self.m, self.v = z(), z(), z(), z()
        self.ema = z() if hp.use_ema else None
        # bf16 mode: a bf16 copy of p at the same element offsets, rewritten by the optimizer kernels; the GEMMs read
        # weights from it (half the bytes, no conversion in the k-loop).  fp32 p stays the master copy.
        self.p16 = torch.zeros(off, dtype=torch.bfloat16, device=device) if shadow_bf16 else None
        block_seg = torch.empty(self.nblocks, dtype=torch.int32)
        preclip, lr_mult, wd, flags = [], [], [], []
        table = spec.group_lr_mult_wd(hp)
        for i, (n, s) in enumerate(segs):
            b0 = self.offset[n] // kk.KK_SEG_ALIGN
            block_seg[b0:b0 + -(-math.prod(s) // kk.KK_SEG_ALIGN)] = i
            if n in spec.param_shapes(dims):
                mx = spec.preclip_max_norm(n, hp)
                preclip.append(mx if mx is not None else 0.0)
                mult, w = table[spec.param_group_of(n)]
                lr_mult.append(mult), wd.append(w)
                flags.append(1 | 2 | (4 if spec.is_weight_norm_target(n) else 0))
            else:                                   # persistent buffer: EMA-tracked only (trainer.py:1504-1517)
                preclip.append(0.0), lr_mult.append(0.0), wd.append(0.0), flags.append(2)
        self.block_seg = block_seg.to(device)
        self.seg_preclip = torch.tensor(preclip, dtype=torch.float32, device=device)
        self.seg_lr_mult = torch.tensor(lr_mult, dtype=torch.float32, device=device)
        self.seg_wd = torch.tensor(wd, dtype=torch.float32, device=device)
        self.seg_flags = torch.tensor(flags, dtype=torch.int32, device=device)
        self.P = {n: self.view(self.p, n) for n in self.names}
        self.G = {n: self.view(self.g, n) for n in self.names}
        self.E = {n: self.view(self.ema, n) for n in self.names} if hp.use_ema else {}
        self.P16 = {n: self.view(self.p16, n) for n in self.names} if shadow_bf16 else {}

    def view(self, slab: torch.Tensor, name: str) -> torch.Tensor:
        o, s = self.offset[name], self.shapes[name]
        return slab[o:o + math.prod(s)].view(s)

    def fused(self, slab: torch.Tensor, first: str, count: int) -> torch.Tensor:
        """[count*H, H] view over `count` consecutive square projection weights (w_q|w_k|w_v are adjacent)."""
        H = self.dims.hidden
        assert (H * H) % kk.KK_SEG_ALIGN == 0
        o = self.offset[first]
        return slab[o:o + count * H * H].view(count * H, H)


def canonical_mel_length(global_mel_length, local_T: int) -> int:
    """The mel length the step's kernels take BY VALUE (kk_losses_finalize, kk_opt_prepare: the batch-shape heuristics of
    trainer.py:2218-2242), canonicalised for the graph keys: both kernels use it only as max(T / 1400, max_dur / 150) > 1, so every
    length up to 1400 frames gives the result of 1400 exactly (T / 1400 <= 1 never decides the max when the max exceeds 1).  Keying the
    step graphs on the raw global length made every (local shape, global T) pair of a ragged data-parallel run a new capture
    (VERDICT r4); with this value a shape is captured once for all global lengths <= 1400 and once per longer length."""
    return max(int(global_mel_length or local_T), 1400)


class KokoroEngine:
    def __init__(self, dims: Optional[ModelDims] = None, hyper: Optional[StepHyper] = None, device="cuda",
                 math_mode: str = "f32", total_steps: int = 20000, seed: int = 0, init: bool = True,
                 storage: str = "auto"):
        kk.load()                                   # fails loudly when libkokoro_hip.so is missing
        if not torch.cuda.is_available():
            raise RuntimeError("KokoroEngine needs an MI355X (no CPU fallback in the product path)")
        self.dims = dims or ModelDims()
        self.dims.validate()
        self.hp = hyper or StepHyper()
        self.device = torch.device(device)
        self.math = {"f32": kk.KK_MATH_F32, "bf16": kk.KK_MATH_BF16}[math_mode]
        self.math_mode = math_mode
        # activation / weight-operand storage: "f32" everywhere (always in the parity mode), or bf16 for the GEMM and
        # attention operands of the bf16 mode ("bf16": both stacks, "bf16-dec": decoder stack only).
        if storage == "auto":
            storage = "bf16" if math_mode == "bf16" else "f32"
        if storage not in ("f32", "bf16", "bf16-dec") or (storage != "f32" and math_mode != "bf16"):
            raise ValueError(f"storage={storage!r} is not available with math_mode={math_mode!r}")
        if storage != "f32" and any(v % 8 for v in (self.dims.enc_ff, self.dims.dec_ff, self.dims.var_filter)):
            raise ValueError("bf16 storage needs enc_ff, dec_ff and var_filter to be multiples of 8")
        self.storage = storage
        self.enc_dt = torch.bfloat16 if storage == "bf16" else torch.float32
        self.dec_dt = torch.bfloat16 if storage in ("bf16", "bf16-dec") else torch.float32
        self.arena = Arena(self.dims, self.hp, self.device, shadow_bf16=storage != "f32")
        self.use_shadow = storage != "f32"
        self.total_steps = total_steps
        # Workspace: ONE flat buffer per name, grown to the largest request ever made under that name, and handed out as
        # views — so a run over ever-changing batch shapes (dynamic batching: B in [4, 32], T up to 1800) holds one
        # activation set sized for the largest batch instead of one per shape.  Growing a buffer moves it, so everything
        # that baked its address in (captured graphs, descriptor tables) is dropped with it (_invalidate).
        self._ws: Dict[str, torch.Tensor] = {}
        self._views: Dict[Tuple, torch.Tensor] = {}
        self.ws_generation = 0
        self.ws_bytes = 0
        self.max_graphs = int(os.environ.get("KK_MAX_GRAPHS", "24"))     # LRU bound of captured batch shapes
        self.max_tables = 8192                                           # descriptor tables (all kinds together)
        self._graphs: "OrderedDict[Tuple, Dict]" = OrderedDict()
        self._tables: "OrderedDict[Tuple, object]" = OrderedDict()
        self._shape_seen: "OrderedDict[Tuple, int]" = OrderedDict()
        # Held while a hipGraph capture is in progress.  Other host threads that talk to the HIP runtime (the trainer's
        # batch prefetcher: event waits, pinned allocations, H2D copies) take it around those calls: on this runtime a
        # concurrent call from another thread invalidates the capture (hipErrorStreamCaptureInvalidated), thread_local
        # capture mode notwithstanding.
        self.capture_lock = threading.Lock()
        self._rope: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._reduce_lists = {"": [], "side.": [], "kv.": []}   # per stream namespace
        self.opt_state = torch.zeros(kk.OS["SIZE"], dtype=torch.float64, device=self.device)
        ns = self.arena.nseg
        f32 = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.seg_gscale, self.seg_decay, self.seg_stepsize = f32(ns), f32(ns), f32(ns)
        self.step_consts = f32(4)
        self.grad_sumsq = torch.zeros(ns, dtype=torch.float64, device=self.device)
        # (Q34.30 fixed-point sums: integer adds commute, so the weight-norm projection decides the same on every replica)
        self.p_sumsq = torch.zeros(ns, dtype=torch.int64, device=self.device)
        # record workspace of kk_seg_sumsq: workgroup partials merged in arena order by a fixed tree (no atomics, no zero-fill)
        self.sumsq_ws = torch.zeros(kk.load().kk_seg_sumsq_ws_bytes(self.arena.nblocks), dtype=torch.uint8, device=self.device)
        self.max_dur = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.loss_acc = torch.zeros(12, dtype=torch.float64, device=self.device)    # 5 sums, 5 counts, non-finite outputs, spare
        self.losses = f32(6)
        self.loss_coef = f32(5)
        self.micro_in_cycle = 0
        self.dp_loss_scale = 1.0                    # 1/world in data-parallel runs (dp.GradSync.loss_scale)
        # Data parallel with ragged shards: an object with .loss_sync(acc[12] f64, max_dur[1] i64) that SUM / MAX all-reduces
        # them in place between the loss forward and the loss backward; the losses are then re-finalised with the global
        # valid-element counts and dp_loss_scale stays 1.  dp.BucketedExchange over RCCL issues the two collectives through
        # the C ABI (kk_comm_loss_sync) on the stream of the step, so they are captured with it: ONE communicator per step,
        # graph replay for ragged shards too.  dp.LossSync (torch.distributed; the gloo tests) is eager only.
        # None = per-rank normalisers (exact for equal shards).
        self.loss_sync = None
        # Data-parallel exchange INSIDE the step (dp.BucketedExchange over the C ABI's kk_comm_*): every bucket of the
        # gradient arena is all-reduced on the exchange's stream as soon as the backward has finished it — after each
        # decoder / encoder layer's grouped weight gradients, the rest after the last launch — and the optimizer waits for
        # that stream.  The collectives are captured with the step, so they are one more branch of its hipGraph (no graph
        # split, nothing between the backward graph and the optimizer graph).  With gradient accumulation only the
        # boundary micro-batch exchanges (_exchange_now).
        self.dp_comm = None
        self._comm_events = {}
        self._exchange_now = True
        self.global_mel_length = None               # batch-max T over all ranks (adaptive loss scale / clip heuristics)
        # dropout / DropPath / SpecAugment: off = the parity configuration (reference with p = 0)
        self.train_dropout = False
        # Second HIP stream: the parts of a step that do not depend on the decoder (pitch/energy predictors; after the
        # loss, the predictors' and the whole text encoder's backward — the length-regulated memory is detached) run
        # beside the decoder kernels instead of after them; they are small launches that cannot fill 256 CUs alone.
        self.overlap = True
        self._side = torch.cuda.Stream(device=self.device)
        self._tmp_ns = ""
        # Third stream: the decoder's input projection and layer-0 self-attention run beside the text encoder (they do not
        # need its output); one fork, one join before the first cross-attention.  Forks are not free in a hipGraph:
        # measured at 8x512, the long independent encoder/predictor branch pays (470K -> 548K frames/s) and so does this
        # one (630K -> 638K), but a per-layer fork for the cross-attention K/V backward lost 9 %, a stream per
        # weight-gradient GEMM 19 %, and moving the duration predictor's forward aside 2 %.
        self._kv = torch.cuda.Stream(device=self.device)
        self.dec_head_aside = True
        # Where the backward's memory tail (all layers' cross-attention K/V weight gradient, the memory gradient, the two
        # bucket-embedding gradients: needs only the cross-attention dK / dV) runs.  0: at the end of the main chain; 3: on the side
        # stream behind the encoder's backward; 4: like 3 with the K/V weight gradient as a member of decoder layer 0's grouped launch;
        # -1: 4 up to 4096 decoder rows, 3 above (measured: profiles/r05_memory_tail_aside_ab.txt).  1 / 2 (a THIRD branch) serialise
        # the side branch behind the main chain in the replayed graph (+21 %): kept for the record of that measurement.
        self.tail_aside = -1
        # bucket-embedding gradients as a segmented sum over frames sorted by bin (kk_bucket_sort beside the predictors' forward)
        self.embed_bwd_sorted = True
        # decoder layer 0's grouped weight-gradient launch on the side stream's idle end instead of the main chain: 1 always, 0 never,
        # -1 where the memory tail is in mode 3 (> 4096 rows: -0.5 ... -1.1 %; at 4096 rows +0.3 ... +1.4 %)
        self.wgrad0_aside = -1
        # legacy schedule (hp.use_onecycle_lr = False): scheduler steps taken so far = epochs completed; the trainer advances it
        self.lr_epoch = 0
        # Fusion switches: plain attributes (tests and tools/probes set them on the object for A/B runs; nothing reads the
        # environment).  Each fused form is tested against the unfused one it replaces.
        self.fuse_glu_fwd = True
        self.group_wgrads = True
        self.fuse_headnorm = True
        self.fuse_headnorm_bwd = True
        # attention backward as ONE launch (kk_attn_bwd: the dQ and the dK/dV kernel as the two halves of a grid), Delta from the
        # epilogue of the w_o dgrad GEMM (kk_gemm_dgrad_delta) — bf16 storage, shapes that take the eight-wave GEMM tile
        self.attn_bwd_pair = True
        # ... or, where the library prices it cheaper (kk_attn_bwd_two_pass: nowhere beside the present pair launch), as the dK/dV
        # kernel that also stores dS + a dQ pass without softmax work (kk_attn_bwd_ws; workspace of 2 bytes per score per stream)
        self.attn_two_pass = True
        # the attention forward stores its dropout keep decisions as packed bits (kk_attn_fwd_kb) and the backward's pair launch reads
        # them (kk_attn_bwd_kb) instead of hashing them again: ~40 % of the backward kernels' vector instructions (round 5)
        self.attn_keep_bits = True
        # ... and, where the persistent encoder launch gives it a place to hide, the bits of ALL the decoder's attention launches come
        # from ONE pure-vector launch beside the encoder forward (kk_attn_keep_gen, on the decoder-head stream) and the forward READS
        # them too (kk_attn_fwd_rb): -3 us per forward launch at 512 frames, -6.4 ... -8.6 us at 1024 (round 6; same bits either way)
        self.attn_keep_gen = True
        # How many decoder layers' bits the generator writes: what fits beside the persistent encoder.  The generator of ALL twelve launches of an
        # 8 x 1024 step outlasts the encoder and delays the decoder head; four layers' worth is the measured optimum there, all six at 8 x 512
        # (profiles/r06_keep_bits_gen_ab.txt; the generator holds 40 registers so that TWO of its waves fit a SIMD beside the encoder's two).
        # Budget = attn_keep_gen_rate 32 x 32 units per microsecond of encoder time (~160 + 2 P us for P phonemes); the later layers' forwards hash and
        # store as before (same bits either way).  0 = never generate.
        self.attn_keep_gen_rate = 1100.0
        # One GPU: the per-segment gradient norms of the weight matrices come from the epilogue of the grouped weight-gradient launches
        # (a record per tile of the FINAL values it stored) instead of from the optimizer's pass over the 199 MB gradient arena, which then
        # reads only what no such launch wrote (embeddings, biases, norms, predictors).  Data parallel keeps the full pass: the norm that
        # clips is the REDUCED gradient's.  _ss_step: the book-keeping of the micro-batch in flight; _ss_ready: what the optimizer boundary
        # behind it may rely on (consumed by optimizer_step; None = full pass).
        self.grad_norm_from_wgrads = True
        self._ss_step = None
        self._ss_ready = None
        self._ss_masks: Dict[frozenset, torch.Tensor] = {}
        self._external_sync = False
        self._seg_of_ptr = {self.arena.G[n].data_ptr(): i for i, n in enumerate(self.arena.names)}
        # Weight warming (kk_attn_warm_next; profiles/r06_l2_retention_probe.txt): an XCD's L2 keeps read-only lines across a kernel
        # boundary, and the attention launches are vector-bound with idle request slots — so the decoder's attention forward touches
        # the lines of the output projection behind it (bit 0) and the dQ half of the backward's pair launch, which ends ~9 us before
        # its dK/dV half, those of the q | k | v (or q) projection whose dgrad follows (bit 2): the GEMMs find their weights L2-hot
        # instead of in HBM.  Interleaved: -0.2 ... -0.6 % at 8 x 512, -0.4 ... -0.7 % at 8 x 1024, level under dynamic batching; bit 1
        # (+ the next sub-layer's first matrix, 0.5 - 3 MB, from the forward) and bit 3 (+ the next output projection from the backward)
        # measured level or worse (profiles/r06_weight_warming_ab.txt).
        self.attn_warm = 5
        self._keep_ready = set()                    # sub-layer keys whose bits kk_attn_keep_gen has written in this step
        # (one-tile sequences — the text encoder's <= 64 phonemes, on the side branch — take the two thinner launches: re-measured INSIDE
        #  the step in round 5, 3.6215 -> 3.608 ms at 8 x 512 (4 of 4 interleaved rounds); 65..128 phonemes keep the pair launch: two launches
        #  there are +0.3 % at 8 x 1024.  profiles/r05_encoder_attn_pair_ab.txt)
        self.attn_pair_min_seq = 64
        self.attn_proj_bf16 = True                 # decoder w_o output stored as bf16 (bf16 mode)
        # the decoder's attention output projection and the sub-layer tail behind it as ONE row-owner launch (kk_linear_tail_fwd) where
        # the library measured it faster (kk_linear_tail_pays: whole rounds of workgroups at >= ~6 K rows); bit-identical results
        self.fuse_linear_tail = True
        # the step's small fp64 accumulators (loss sums, per-segment gradient / parameter norms) are kept zero by their last readers
        # instead of a zero-fill launch in front of every writer (three dependent launches of the critical chain)
        # _acc_clean: host-side record of the invariant "the accumulators are zero between steps".  It is dropped while a writer ..
        # cleaner sequence is in flight and restored only when the sequence has been issued completely (_acc_guard): a step that
        # aborts in between (a collective or a launch raising), or a flip of the switch on a live engine, costs one zero-fill
        # instead of silently adding onto stale sums.
        self._acc_clean = True
        self._acc_depth = 0
        self._self_cleaning_acc = True
        # key-padding mask + embedding (+ PE, dropout) + the first encoder layer's pre-LayerNorm as one launch (kk_embed_ln_fwd): the
        # head of the critical path in front of the encoder forward is 3 dependent launches instead of 5
        self.fuse_enc_prologue = True
        # kk_rowdot_bwd (the predictors' / stop head's Linear(C -> 1) backward) leaves its weight-gradient sums as one plain row per
        # workgroup for the backward's single kk_partials_reduce instead of 256 workgroups' atomics on the same C + 1 addresses
        self.rowdot_partials = True
        # length-regulator gather + pitch / energy embedding adds + SpecAugment of the memory as one launch (kk_regulate_embed_fwd)
        self.fuse_memory_fwd = True
        # The zero-fill at the start of an accumulation cycle skips what the cycle's first grouped weight-gradient launches
        # overwrite (89 % of the arena at default dims; the fill runs beside the latency-bound encoder launch: 20 us of the step).
        # Which tensors those are is RECORDED from the launches of a step (per precision mode), never assumed, and a step that
        # zeroed by the record checks at its end that it overwrote exactly that set (_zero_grad_step / _grouped_wgrads).
        self.zero_skip_overwritten = True
        self._ow_sets: Dict[Tuple, frozenset] = {}
        self._ow_seen: Optional[set] = None
        self._ow_expect: Optional[frozenset] = None
        self._wgrad_queue = {}
        self._defer_wgrads = None                   # a list while a caller collects grouped weight-gradient launches for later
        # The gradient arena was zeroed for THIS micro-batch (first of an accumulation cycle): a layer's grouped weight
        # gradients are each written exactly once per micro-batch, so they overwrite instead of read-modify-write (dW is
        # 31 MB per decoder layer: a sixth of the grouped launch's traffic).  _first_micro: set by train_step around its call.
        self._grads_fresh = False
        self._first_micro = False
        # Text-encoder forward as one persistent launch (kk_encoder_stack_fwd): bf16 mode, phoneme sequences <= 128;
        # otherwise (and with KK_ENC_FUSED=0) the per-kernel sequence.  _enc_sync: its group-barrier words (word 0 != 0
        # = a barrier timed out; encoder_stack_error() reads it).
        self.enc_fused = True
        self.enc_fused_max_batch = 8                # one item per workgroup group (see _encoder_stack_ok)
        self.zero_late = True                       # gradient zero-fill after the decoder head's first launches
        self.enc_placement = 0                      # (tests force the groups across XCDs with 1)
        self._enc_sync = torch.zeros(512, dtype=torch.int32, device=self.device)
        self.enc_trace, self.enc_trace_wg = None, 0      # tools/probes/enc_stack_phases.py: per-phase clock stamps of one workgroup
        self.spec_augment_active = True             # the trainer clears it for epochs < spec_augment_start_epoch
        # KK_TRACE=1: one-thread time-stamp launches at the marks of a step (also inside the captured graphs), read back by
        # timeline() — the real overlap of the graph's branches, which rocprofv3 cannot show (it serialises them)
        self.trace = os.environ.get("KK_TRACE", "0") == "1"
        self._marks: Dict[str, int] = {}
        self._mark_buf = torch.zeros(512, dtype=torch.int64, device=self.device) if self.trace else None
        self.rng = torch.full((1,), int(seed) & 0x7FFFFFFF, dtype=torch.int32, device=self.device)   # step seed, read on device
        for n, b in spec.make_buffers(self.dims).items():
            self.arena.P[n].copy_(b)
        if init:
            self.load_params(spec.init_params(self.dims, seed))
        elif self.arena.ema is not None:
            self.arena.ema.copy_(self.arena.p)
        self.sync_shadow()

    @property
    def self_cleaning_acc(self) -> bool:
        return self._self_cleaning_acc

    @self_cleaning_acc.setter
    def self_cleaning_acc(self, on) -> None:
        if bool(on) != self._self_cleaning_acc:     # (the other mode's last reader did not leave the sums as this mode expects them)
            self._acc_clean = False
        self._self_cleaning_acc = bool(on)

    @contextlib.contextmanager
    def _acc_guard(self):
        """Around every sequence that writes and then cleans the handed-round accumulators (loss_acc in _fb; p_sumsq in
        optimizer_step; a graph capture / replay of either): restores the zero state first when an earlier sequence did not complete."""
        if self._acc_depth == 0:
            if not self._acc_clean:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("the step's accumulators are not in their zero state inside a graph capture")
                self.loss_acc.zero_()
                self.grad_sumsq.zero_()
                self.p_sumsq.zero_()
            self._acc_clean = False
        self._acc_depth += 1
        try:
            yield
        except BaseException:
            self._acc_depth -= 1
            raise
        self._acc_depth -= 1
        if self._acc_depth == 0:
            self._acc_clean = True

    # ------------------------------------------------------------------ state
    def load_params(self, params: Dict[str, torch.Tensor], reset_ema: bool = True) -> None:
        for n in self.arena.param_names:
            self.arena.P[n].copy_(params[n].to(self.device, torch.float32).view(self.arena.shapes[n]))
        if reset_ema and self.arena.ema is not None:
            self.arena.ema.copy_(self.arena.p)      # EMA starts as a deep copy of the model (trainer.py:835)
        self.sync_shadow()

    @contextlib.contextmanager
    def fp32_math(self):
        """Run the enclosed calls in the fp32 parity mode (fp32 MFMA, fp32 storage, master weights) whatever the
        engine's training precision is — the reference validates without autocast (trainer.py:1821-1834)."""
        saved = (self.math, self.enc_dt, self.dec_dt, self.use_shadow)
        self.math, self.enc_dt, self.dec_dt, self.use_shadow = kk.KK_MATH_F32, torch.float32, torch.float32, False
        try:
            yield self
        finally:
            self.math, self.enc_dt, self.dec_dt, self.use_shadow = saved

    @contextlib.contextmanager
    def ema_weights(self):
        """Run the enclosed forward passes on the EMA replica (reference validation, trainer.py:1771-1790) by pointing the
        parameter views at the EMA slab — no copy, and nothing but two references to restore if the block raises.  Only
        meaningful without the bf16 weight shadow (combine with fp32_math())."""
        a = self.arena
        if a.ema is None:
            yield self
            return
        if self.use_shadow:
            raise RuntimeError("ema_weights(): the bf16 shadow holds the live weights; use it inside fp32_math()")
        saved = (a.P, a.p)
        a.P, a.p = a.E, a.ema
        try:
            yield self
        finally:
            a.P, a.p = saved

    def sync_shadow(self) -> None:
        """Rebuild the bf16 weight shadow from the fp32 master arena (after any write to arena.p from outside the
        optimizer kernels)."""
        if self.arena.p16 is not None:
            kk.call("kk_cast_f32_bf16", self.arena.p, self.arena.p16, self.arena.total)

    def state_dict(self, ema: bool = False) -> "OrderedDict[str, torch.Tensor]":
        """311 reference-named tensors (views of the arena; clone before mutating)."""
        src = self.arena.E if ema else self.arena.P
        return OrderedDict((n, src[n]) for n in spec.state_dict_order(self.dims))

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        want = set(spec.state_dict_order(self.dims))
        missing, unexpected = want - set(sd), set(sd) - want
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing={sorted(missing)} unexpected={sorted(unexpected)}")
        for n in want & set(sd):
            if tuple(sd[n].shape) != tuple(self.arena.shapes[n]):
                raise RuntimeError(f"size mismatch for {n}: {tuple(sd[n].shape)} vs {self.arena.shapes[n]}")
            self.arena.P[n].copy_(sd[n].to(self.device, torch.float32))
        self.sync_shadow()

    def grads(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((n, self.arena.G[n]) for n in self.arena.param_names)

    # ------------------------------------------------------------------ helpers
    @contextlib.contextmanager
    def _on_stream(self, stream, ns, enable=True, after=None):
        """Run the enclosed launches on `stream`, after everything already queued on the current stream (or, with
        `after`, after that earlier point of it: an event from _fork_point); the caller joins with _join(stream).
        Scratch ("tmp.*") buffers get the namespace `ns` so that streams never share one.
        With overlap off this is a no-op (same stream, same order).

        Under rocprofv3 the branch captured FIRST after a fork runs first and the other one starts late (the decoder
        backward appeared 1.1 ms after the losses when the encoder backward was captured before it), so the
        critical-path branch is captured first and the side branch afterwards, forked from an event recorded at the
        fork point.  Unprofiled, the step time is the same either way."""
        if not (self.overlap and enable):
            yield
            return
        saved, self._tmp_ns = self._tmp_ns, ns
        try:
            if after is not None:
                stream.wait_event(after)
            else:
                stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                yield
        finally:
            self._tmp_ns = saved

    def _on_side_stream(self, after=None):
        return self._on_stream(self._side, "side.", after=after)

    def _fork_point(self):
        """An event on the current stream that a later _on_stream(..., after=event) forks from."""
        if not self.overlap:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def _comm_bucket(self, tag: str) -> None:
        """The gradients of bucket `tag` are final on the current stream from here on: exchange them on the comm stream."""
        c = self.dp_comm
        if c is None or not self._exchange_now:
            return
        # buckets travel in GROUPS (dp.BucketedExchange.groups): the bucket's event is recorded where it becomes final; when the last
        # bucket of a group has arrived the communication branch waits for all of the group's events and carries ONE exchange
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._comm_events.setdefault(c._group_of[tag], []).append(ev)
        tags = c.arrive(tag)
        if tags is None:
            return
        events = self._comm_events.pop(c._group_of[tag])
        for e in events:
            c.stream.wait_event(e)
        with torch.cuda.stream(c.stream):
            c.reduce_tags(self.arena.g, tags)

    def _comm_join(self) -> None:
        c = self.dp_comm
        if c is not None and self._exchange_now:
            torch.cuda.current_stream().wait_stream(c.stream)

    def _mark(self, name: str) -> None:
        if self.trace:
            idx = self._marks.setdefault(name, len(self._marks))
            kk.call("kk_timestamp", self._mark_buf[idx:])

    def timeline(self):
        """[(microseconds since the step's first mark, name)] of the last executed step (KK_TRACE=1), sorted by time."""
        if not self.trace:
            raise RuntimeError("timeline(): set KK_TRACE=1 before constructing the engine")
        t = self._mark_buf.cpu().tolist()
        rows = sorted((t[i], n) for n, i in self._marks.items() if t[i])
        return [((v - rows[0][0]) / 100.0, n) for v, n in rows]          # 100 MHz ticks

    def _join(self, stream) -> None:
        if self.overlap:
            torch.cuda.current_stream().wait_stream(stream)

    def _join_side(self) -> None:
        self._join(self._side)

    def _buf(self, key, *shape, dtype=torch.float32) -> torch.Tensor:
        """A [shape] view of the workspace buffer `key` (scratch names "tmp.*" are private to the current stream)."""
        if key.startswith("tmp."):
            key = self._tmp_ns + key
        vk = (key, shape, dtype)
        v = self._views.get(vk)
        if v is not None:
            return v
        need = math.prod(shape) * _ITEMSIZE[dtype]
        t = self._ws.get(key)
        if t is None or t.numel() < need:
            t = self._grow(key, need)
        v = t[:need].view(dtype).view(shape)
        if len(self._views) > 32768:
            self._views.clear()
        self._views[vk] = v
        return v

    def _grow(self, key: str, need: int) -> torch.Tensor:
        old = self._ws.get(key)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"workspace '{key}' would be allocated inside a graph capture (run the shape eagerly first)")
        if old is not None:
            # the old block may still be in use by queued kernels of any stream, and captured graphs / tables hold its
            # address: drain, drop them, then let it go.  12.5 % headroom so that slowly creeping shapes do not regrow
            # every step.
            torch.cuda.synchronize(self.device)
            self._invalidate()
            self.ws_bytes -= old.numel()
            need = need + need // 8
        need = (need + 255) // 256 * 256
        t = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ws[key] = t
        self.ws_bytes += need
        return t

    def _invalidate(self) -> None:
        """Forget everything that holds workspace addresses: views, descriptor tables, captured graphs."""
        self._views.clear()
        self._tables.clear()
        self._graphs.clear()
        self.ws_generation += 1

    def _table(self, key: Tuple, build):
        """Descriptor table cache (pointer / weight-gradient / head-norm / reduction tables), LRU-bounded.  A captured graph
        reads device-resident tables when it replays, so evicting tables also drops the graphs."""
        t = self._tables.get(key)
        if t is None:
            if len(self._tables) >= self.max_tables:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("descriptor-table cache overflow inside a graph capture")
                torch.cuda.synchronize(self.device)
                self._graphs.clear()
                for _ in range(self.max_tables // 4):
                    self._tables.popitem(last=False)
            t = self._tables[key] = build()
        else:
            self._tables.move_to_end(key)
        return t

    def workspace_bytes(self) -> int:
        """Bytes held by the activation workspace (bounded by the largest batch seen, not by the number of shapes)."""
        return self.ws_bytes

    def _rope_tables(self, S: int):
        """cos / sin tables for positions [0, S): prefixes of one pair built once for the longest supported sequence
        (2 x max_len x 64 floats), so their addresses never change under captured graphs."""
        if not self._rope:
            c, s = spec.rope_tables(self.dims.max_len, 64)
            self._rope[0] = (c.to(self.device), s.to(self.device))
        if S > self.dims.max_len:
            raise ValueError(f"sequence of {S} positions exceeds the positional table ({self.dims.max_len})")
        c, s = self._rope[0]
        return c[:S], s[:S]

    def _W(self, name: str) -> torch.Tensor:
        """GEMM weight operand: the bf16 shadow when there is one, else the fp32 master."""
        if self.use_shadow and self.arena.shapes[name][-1] % 8 == 0:      # bf16 rows are fetched 8 elements at a time
            return self.arena.P16[name]
        return self.arena.P[name]

    def _Wconv(self, name: str, rows: int, cols: int) -> torch.Tensor:
        """Conv1d(k=3) weight [C_out, C_in, 3] as the [C_out, 3*C_in] operand of the unfolded GEMM: the bf16 shadow when the
        row length allows 16-byte fetches (the shape's last dim is 3, so _W would fall back to the fp32 master)."""
        if self.use_shadow and cols % 8 == 0:
            return self.arena.P16[name].view(rows, cols)
        return self.arena.P[name].view(rows, cols)

    def _Wf(self, first: str, count: int) -> torch.Tensor:
        a = self.arena
        return a.fused(a.p16 if self.use_shadow else a.p, first, count)

    def _linear(self, x, W, b, out, res=None, res_mod=0):
        N, K = x.shape
        M = W.shape[0]
        # split_k = 0: kk_gemm splits K (fp32 atomics) by itself when the tile grid is too small to fill the chip
        kk.call("kk_gemm", 0, 0, N, M, K, 1.0, x, x.stride(0), W, K, 0.0, out, out.stride(0), b, res,
                res.stride(0) if res is not None else 0, res_mod, 0, self.math, _b16(x) | _b16(W) << 1 | _b16(out) << 2)

    def _dgrad(self, dy, W, dx, beta=0.0):
        N, M = dy.shape
        K = W.shape[1]
        kk.call("kk_gemm", 0, 1, N, K, M, 1.0, dy, dy.stride(0), W, K, beta, dx, dx.stride(0), None, None, 0, 0, 0, self.math,
                _b16(dy) | _b16(W) << 1 | _b16(dx) << 2)

    def _wgrad(self, dy, x, dW, db=None):
        N, M = dy.shape
        K = x.shape[1]
        q = self._wgrad_queue.get(self._tmp_ns)
        if q is not None and db is None and _b16(dy) and _b16(x) and self.math == kk.KK_MATH_BF16:
            q.append((dy, x, dW))                        # issued with the rest of the layer's weight gradients
            return
        kk.call("kk_gemm", 1, 1, M, K, N, 1.0, dy, dy.stride(0), x, x.stride(0), 1.0, dW, K, None, None, 0, 0, 0, self.math,
                _b16(dy) | _b16(x) << 1)
        if db is not None:
            kk.call("kk_colsum_acc", dy, dy.stride(0), N, M, db, _b16(dy))

    @contextlib.contextmanager
    def _grouped_wgrads(self):
        """Weight-gradient GEMMs issued inside the block (one layer's backward) are queued and run as ONE grouped launch
        at its end: nothing reads a weight gradient before the optimizer, and together they fill the chip with
        full-length reductions (no split-K atomics).  Their operands — the saved activations and the layer's own
        gradient buffers — are not rewritten before the next layer's backward starts."""
        ns = self._tmp_ns
        if not self.group_wgrads or ns in self._wgrad_queue:
            yield
            return
        q = self._wgrad_queue[ns] = []
        try:
            yield
        finally:
            del self._wgrad_queue[ns]
        if self._defer_wgrads is not None:               # (the caller issues this group later, elsewhere: _issue_wgrad_group)
            self._defer_wgrads.append(q)
            return
        self._issue_wgrad_group(q)

    def _issue_wgrad_group(self, q) -> None:
        for i in range(0, len(q), 8):
            part = q[i:i + 8]
            sig = ("wg",) + tuple((dy.data_ptr(), x.data_ptr(), dw.data_ptr(), tuple(dy.shape), tuple(x.shape)) for dy, x, dw in part)
            table = self._table(sig, lambda: kk.wgrad_table(part))
            st = self._ss_step
            if st is None:
                kk.call("kk_gemm_wgrad_group", table, len(part), 0, 1 if self._grads_fresh else 0, None, None, None)
            else:
                # (one GPU) the tiles of a launch that writes every element once also leave the sums of squares of what they stored as
                # records of kk_seg_sumsq's workspace: the optimizer's norm pass then skips those tensors (the library decides per launch
                # and says so by advancing the count)
                segs, flat = [], []
                for _, _, dw in part:
                    si = self._seg_of_ptr.get(dw.data_ptr())
                    if si is None:
                        segs = None
                        break
                    rows = self.arena.shapes[self.arena.names[si]][0]
                    cnt = dw.shape[0] // rows                  # (fused q|k|v / k|v views: `cnt` adjacent segments of `rows` rows each)
                    if dw.shape[0] % rows or any(self.arena.shapes[self.arena.names[si + c]] != self.arena.shapes[self.arena.names[si]] for c in range(cnt)):
                        segs = None
                        break
                    segs.extend(range(si, si + cnt))
                    flat.extend((si, rows if cnt > 1 else 0))
                n0 = st["count"].value
                if segs is None or n0 + 1024 > st["cap"]:
                    kk.call("kk_gemm_wgrad_group", table, len(part), 0, 1 if self._grads_fresh else 0, None, None, None)
                else:
                    seg_arr = (kk.C.c_int32 * len(flat))(*flat)
                    kk.call("kk_gemm_wgrad_group", table, len(part), 0, 1 if self._grads_fresh else 0, st["rec"], seg_arr, kk.C.byref(st["count"]))
                    if st["count"].value != n0:
                        st["covered"].update(segs)
            if self._grads_fresh and self._ow_seen is not None:
                self._ow_seen.update((dw.data_ptr(), dw.numel()) for _, _, dw in part)

    def _ln_fwd(self, key, x, prefix, dtype=torch.float32):
        P = self.arena.P
        rows, H = x.shape
        y, mean, rstd = self._buf(key + ".y", rows, H, dtype=dtype), self._buf(key + ".mean", rows), self._buf(key + ".rstd", rows)
        kk.call("kk_layernorm_fwd", x, P[prefix + ".weight"], P[prefix + ".bias"], y, mean, rstd, rows, H, _b16(y))
        return y

    def _partials(self, key, rows, H, ncols, dst0, dst1, split):
        """Per-workgroup column partial sums of a norm backward, summed into the gradient vectors by the single
        kk_partials_reduce launch at the end of the backward pass (instead of device-scope atomics per workgroup)."""
        nb = kk.load().kk_norm_bwd_blocks(rows, H)
        part = self._buf(key + ".part", nb, ncols)
        self._reduce_lists[self._tmp_ns].append((part, dst0, dst1, nb, ncols, split))
        return part

    def _rowdot_partials(self, key, rows, C, dw, db):
        """[blocks][C + 4] partial (dw | db) rows of one kk_rowdot_bwd launch, summed by the backward's kk_partials_reduce — or None
        (same-address atomics inside the launch) when C is not a multiple of 4."""
        if C % 4 or not self.rowdot_partials:
            return None
        nb = kk.load().kk_rowdot_bwd_blocks(rows)
        part = self._buf(key + ".rdpart", nb, C + 4)
        self._reduce_lists[self._tmp_ns].append((part, dw.view(-1), db.view(-1), nb, C + 1, C, C + 4))
        return part

    def _headnorm_partials(self, key, rows, dgains):
        """[parts][blocks][64] partial gain gradients of one head-norm backward launch, one reduce descriptor per part."""
        nb = kk.load().kk_headnorm_bwd_blocks(rows, self.dims.heads)
        part = self._buf(key + ".hpart", len(dgains), nb, 64)
        for j, dg in enumerate(dgains):
            self._reduce_lists[self._tmp_ns].append((part[j], dg, None, nb, 64, 64))
        return part

    def _ln_bwd(self, key, dy, x, prefix, dx, accumulate):
        P, G = self.arena.P, self.arena.G
        rows, H = x.shape
        part = self._partials(key, rows, H, 2 * H, G[prefix + ".weight"], G[prefix + ".bias"], H)
        kk.call("kk_layernorm_bwd", dy, x, P[prefix + ".weight"], self._buf(key + ".mean", rows), self._buf(key + ".rstd", rows),
                dx, 1 if accumulate else 0, G[prefix + ".weight"], G[prefix + ".bias"], part, rows, H, _b16(dy))

    def _tail_bwd(self, key, dn, x, prefix, dres, accumulate, head) -> None:
        """Backward of LayerNorm `key` (input x = the residual stream after the sub-layer `head`), fused — when that
        sub-layer ran its dropout tail — with the head of the sub-layer's own backward (kk_sublayer_in_bwd): masks,
        RMSNorm backward for an FFN, bias column sums.  head = (kind, ffn_key, prefix, S, site, p, dpr, dtype).
        The sub-layer's backward then starts from the prepared buffers (tmp.df2 / tmp.d_attn_proj)."""
        kind, hkey, hprefix, S, site, p, dpr, dt = head
        P, G = self.arena.P, self.arena.G
        rows, H = x.shape
        nb = kk.load().kk_sublayer_in_bwd_blocks(rows)
        part = self._buf(key + ".tpart", nb, 4 * H)
        self._reduce_lists[self._tmp_ns].append((part, G[prefix + ".weight"], G[prefix + ".bias"], nb, 2 * H, H, 4 * H))
        if kind == "ffn":
            f2, gain = self._buf(hkey + ".f2", rows, H, dtype=dt), P[hprefix + ".output_norm.weight"]
            rstd_f, dy = self._buf(hkey + ".rstd_f", rows), self._buf("tmp.df2", rows, H, dtype=dt)
            self._reduce_lists[self._tmp_ns].append((part[:, 2 * H:], G[hprefix + ".linear2.bias"], G[hprefix + ".output_norm.weight"], nb, 2 * H, H, 4 * H))
            p2 = p
        else:
            f2 = gain = rstd_f = None
            dy = self._buf("tmp.d_attn_proj" + (".x" if hprefix.endswith(".cross_attn") else ""), rows, H, dtype=dt)
            self._reduce_lists[self._tmp_ns].append((part[:, 2 * H:], G[hprefix + ".w_o.bias"], None, nb, H, H, 4 * H))
            p2 = 0.0
        kk.call("kk_sublayer_in_bwd", dn, _b16(dn), x, P[prefix + ".weight"], self._buf(key + ".mean", rows), self._buf(key + ".rstd", rows),
                dres, 1 if accumulate else 0, f2, gain, rstd_f, dy, _b16(dy), part, rows, H, S, self.rng, site, p, site + 1, p2,
                site + 2, dpr)

    def _reduce_partials(self, shape_key) -> None:
        """One launch that adds the column sums of every partial matrix written so far (by streams already joined into
        the current one) to its gradient vectors."""
        todo = [e for ns in ("side.", "kv.", "") for e in self._reduce_lists[ns]]
        for ns in self._reduce_lists:
            self._reduce_lists[ns] = []
        if not todo:
            return
        tkey = ("red", shape_key, tuple((e[0].data_ptr(), e[3]) for e in todo))      # which partial matrices this pass wrote
        # workspace addresses are stable while no buffer grows: the table is built once per batch shape
        ent = self._table(tkey, lambda: (kk.reduce_table(todo, self.device), len(todo), max(e[4] for e in todo)))
        kk.call("kk_partials_reduce", ent[0], ent[1], ent[2])

    # ------------------------------------------------------------------ dropout plumbing
    def _p(self, rate: float) -> float:
        return float(rate) if self.train_dropout else 0.0

    def _dpr(self, i: int, n: int) -> float:
        """Stochastic-depth rate of layer i of n (model.py:100-107)."""
        if not (self.train_dropout and self.hp.use_stochastic_depth):
            return 0.0
        return (i / max(n - 1, 1)) * self.hp.stochastic_depth_rate

    def _sublayer_tail(self, y, x_res, x_out, S, site, p, dpr, p2, gain, rstd_f, next_ln):
        """Fused tail of a sub-layer on the dropout path (kk_sublayer_out_fwd): x_out = x_res + masks * [RMSNorm](y), and
        the LayerNorm `next_ln = (key, prefix, dtype)` of x_out when the caller names one; returns that LayerNorm's
        output (else None)."""
        P = self.arena.P
        rows, H = y.shape
        n = mean = rstd = g = b = None
        if next_ln is not None:
            key, prefix, dtype = next_ln
            n, mean, rstd = self._buf(key + ".y", rows, H, dtype=dtype), self._buf(key + ".mean", rows), self._buf(key + ".rstd", rows)
            g, b = P[prefix + ".weight"], P[prefix + ".bias"]
        kk.call("kk_sublayer_out_fwd", y, _b16(y), gain, rstd_f, x_res, x_out, g, b, n, _b16(n), mean, rstd, rows, H, S, self.rng,
                site, p, site + 1, p2, site + 2, dpr)
        return n

    # ------------------------------------------------------------------ attention sub-layer
    def _attn_fwd(self, key, prefix, xq, xkv, B, Sq, Sk, rope, causal, key_mask, x_res, x_out, site=0, p=0.0, dpr=0.0, next_ln=None,
                  layer=0):
        """x_out = x_res + w_o(attention(...)) + b_o.  xq [B*Sq,H] (post-LN), xkv [B*Sk,H] (None = self-attention).
        Returns LayerNorm_next_ln(x_out) when the fused dropout tail computed it, else None."""
        P, H, h = self.arena.P, self.dims.hidden, self.dims.heads
        Nq, Nk = B * Sq, B * Sk
        dt = xq.dtype                                   # storage of every activation of the sub-layer
        i16 = _b16(xq)
        cos, sin = self._rope_tables(max(Sq, Sk)) if rope else (None, None)
        gq, gk, gv = P[prefix + ".q_norm.weight"], P[prefix + ".k_norm.weight"], P[prefix + ".v_norm.weight"]
        if xkv is None:       # q|k|v from one fused projection; RoPE on q and k only
            raw, nrm = self._buf(key + ".qkv_raw", Nq, 3 * H, dtype=dt), self._buf(key + ".qkv_n", Nq, 3 * H, dtype=dt)
            self._proj_headnorm(xq, self._Wf(prefix + ".w_q.weight", 3), raw, nrm, Sq, (gq, gk, gv), 3 if rope else 0, cos, sin)
            q_raw, k_raw, v_raw, q_n, k_n, v_n = raw, raw[:, H:], raw[:, 2 * H:], nrm, nrm[:, H:], nrm[:, 2 * H:]
        else:
            q_raw, q_n = self._buf(key + ".q_raw", Nq, H, dtype=dt), self._buf(key + ".q_n", Nq, H, dtype=dt)
            kv_raw, kv_n = self._cross_kv(layer, Nk, dt)                              # filled by _cross_kv_fwd_all
            self._proj_headnorm(xq, self._W(prefix + ".w_q.weight"), q_raw, q_n, Sq, (gq,), 0, None, None)
            k_raw, v_raw, k_n, v_n = kv_raw, kv_raw[:, H:], kv_n, kv_n[:, H:]
        ctx, lse = self._buf(key + ".ctx", Nq, H, dtype=dt), self._buf(key + ".lse", B, h, Sq)
        keep = self._attn_keep(key, B, Sq, Sk, p, i16)
        if (self.attn_warm & 3) and i16 and key.startswith("dec") and self.use_shadow:
            # the forward warms the weights of the GEMMs behind it into every XCD's L2 (kk_attn_warm_next): its own output projection, and
            # the next sub-layer's first matrix (cross-attention: w_q; feed-forward: linear1)
            base = prefix.rsplit(".", 1)[0]                # "decoder.layers.<i>"
            nxt = base + (".cross_attn.w_q.weight" if xkv is None else ".ff.linear1.weight")
            w0, w1 = self._W(prefix + ".w_o.weight"), self._W(nxt)
            kk.load().kk_attn_warm_next(w0.data_ptr(), w0.numel() * w0.element_size(), w1.data_ptr() if self.attn_warm & 2 else None,
                                        w1.numel() * w1.element_size())
        if keep is not None and key in self._keep_ready:      # the bits are there already (kk_attn_keep_gen): the forward reads them
            kk.call("kk_attn_fwd_rb", q_n, k_n, v_n, ctx, lse, B, h, Sq, Sk, q_n.stride(0), k_n.stride(0), v_n.stride(0), H, key_mask,
                    1 if causal else 0, 0.125, self.rng, site + 3, p, self.math, i16, keep)
        elif keep is not None:    # the forward also stores the dropout keep decisions (1 bit per score) for the backward's pair launch
            kk.call("kk_attn_fwd_kb", q_n, k_n, v_n, ctx, lse, B, h, Sq, Sk, q_n.stride(0), k_n.stride(0), v_n.stride(0), H, key_mask,
                    1 if causal else 0, 0.125, self.rng, site + 3, p, self.math, i16, keep)
        else:
            kk.call("kk_attn_fwd", q_n, k_n, v_n, ctx, lse, B, h, Sq, Sk, q_n.stride(0), k_n.stride(0), v_n.stride(0), H, key_mask,
                    1 if causal else 0, 0.125, self.rng, site + 3, p, self.math, i16)
        # the projection output lives only until the tail two launches later: in the decoder's bf16 mode it is stored like every
        # other GEMM result there (what autocast gives the reference's nn.Linear); the text encoder keeps fp32 (its persistent
        # launch hands the tile over in fp32, and the per-kernel path must match it)
        Wo = self._W(prefix + ".w_o.weight")
        p16 = self.attn_proj_bf16 and i16 and key.startswith("dec")
        if (p16 and self.fuse_linear_tail and _b16(Wo) and self.math == kk.KK_MATH_BF16 and kk.load().kk_linear_tail_pays(Nq, H, H)):
            # projection + tail as ONE row-owner launch (kk_linear_tail_fwd: same bits as the two launches below)
            n = mean = rstd = g = b = None
            if next_ln is not None:
                lkey, lprefix, ldt = next_ln
                n, mean, rstd = self._buf(lkey + ".y", Nq, H, dtype=ldt), self._buf(lkey + ".mean", Nq), self._buf(lkey + ".rstd", Nq)
                g, b = P[lprefix + ".weight"], P[lprefix + ".bias"]
            kk.call("kk_linear_tail_fwd", ctx, ctx.stride(0), Wo, P[prefix + ".w_o.bias"], H, None, 1, None, None, x_res, x_out, g, b, n,
                    _b16(n), mean, rstd, Nq, H, Sq, self.rng, site, p, site + 1, 0.0, site + 2, dpr)
            return n
        proj = self._buf("tmp.attn_proj16" if p16 else "tmp.attn_proj", Nq, H, dtype=dt if p16 else torch.float32)
        self._linear(ctx, Wo, P[prefix + ".w_o.bias"], proj)
        return self._sublayer_tail(proj, x_res, x_out, Sq, site, p, dpr, 0.0, None, None, next_ln)   # (p = 0: masks are all ones)

    def _attn_keep(self, key, B, Sq, Sk, p, i16):
        """Buffer of the attention sub-layer's packed dropout keep decisions (kk_attn_fwd_kb writes, kk_attn_bwd_kb reads: the backward
        then spends no vector instructions on the mask hash), or None where the library stores none (short sequences, fp32 storage,
        no dropout) or the switch is off."""
        if not (self.attn_keep_bits and i16 and p > 0.0):
            return None
        n = kk.load().kk_attn_keep_bytes(B, self.dims.heads, Sq, Sk)
        return self._buf(key + ".keep", n, dtype=torch.uint8) if n > 0 else None

    def _keep_gen_launch(self, B, T, p_dec, seed_offset=0, mark_ready=True, max_wgs=0, layers=None) -> bool:
        """kk_attn_keep_gen for every decoder attention launch of a (B, T) step; False when that shape has no keep-bit arrays."""
        d, ents = self.dims, []
        for li in range(d.dec_layers if layers is None else min(layers, d.dec_layers)):
            for sub, off, cz in ((".sa", 0, True), (".ca", 8, False)):
                kb = self._attn_keep(f"dec{li}{sub}", B, T, T, p_dec, 1)
                if kb is not None:
                    ents.append((kb, 2000 + 32 * li + off + 3, p_dec, B, d.heads, T, T, cz))
                    if mark_ready:
                        self._keep_ready.add(f"dec{li}{sub}")
        for c0 in range(0, len(ents), 16):
            part = ents[c0:c0 + 16]
            table = self._table(("keepgen",) + tuple((e[0].data_ptr(), e[1], e[2], B, T) for e in part), lambda part=part: kk.keep_sites(part))
            kk.call("kk_attn_keep_gen", table, len(part), self.rng, seed_offset, max_wgs)
        return bool(ents)

    def _cross_kv(self, layer, Nk, dt, which=""):
        """(raw, normed) K|V of cross-attention layer `layer`: column slices [.., 2H] of the all-layer buffers (row stride
        layers*2H); which = "d" for their gradients."""
        H, L = self.dims.hidden, self.dims.dec_layers
        raw = self._buf(f"dec.ca.{which}kv_raw_all", Nk, 2 * H * L, dtype=dt)
        nrm = self._buf(f"dec.ca.{which}kv_n_all", Nk, 2 * H * L, dtype=dt)
        return raw[:, 2 * H * layer:2 * H * (layer + 1)], nrm[:, 2 * H * layer:2 * H * (layer + 1)]

    def _proj_headnorm(self, x, W, raw, nrm, S, gains, rope_mask, cos, sin):
        """raw = x.W^T (an attention projection of len(gains) parts x heads x 64 columns, no bias: transformers.py:131-136)
        and nrm = its per-head RMSNorm (+ RoPE): in bf16 mode one GEMM whose epilogue normalises, else GEMM + norm launches."""
        H, h = self.dims.hidden, self.dims.heads
        rows, parts = x.shape[0], len(gains)
        if self.fuse_headnorm and _b16(x) and _b16(W) and _b16(raw) and parts <= 12 and x.shape[1] % 64 == 0:
            table = self._table(("ptr",) + tuple(g.data_ptr() for g in gains), lambda: kk.pointer_table(gains))
            kk.call("kk_gemm_qkv_headnorm", rows, parts, h, x.shape[1], x, x.stride(0), W, None, raw, raw.stride(0), nrm,
                    nrm.stride(0), S, table, rope_mask, cos, sin)
            return
        self._linear(x, W, None, raw)
        for p0 in range(0, parts, 3):                      # the norm kernel takes up to three parts per launch
            g = list(gains[p0:p0 + 3]) + [None] * 3
            n = min(3, parts - p0)
            kk.call("kk_headnorm_rope_fwd", raw[:, p0 * H:], raw.stride(0), nrm[:, p0 * H:], nrm.stride(0), rows, h, S, n,
                    g[0], g[1], g[2], (rope_mask >> p0) & 7, cos, sin, _b16(raw))

    def _cross_kv_fwd_all(self, xkv, Nk, Sk, dt, first=0, last=None):
        """K/V projections of ALL decoder cross-attention layers in one GEMM (they depend on the memory alone and their
        weights are contiguous in the arena), then the per-head RMSNorm of each layer's slice (no RoPE:
        transformers.py:268-277 applies it to self-attention only)."""
        P, H, h, L = self.arena.P, self.dims.hidden, self.dims.heads, self.dims.dec_layers
        raw_all = self._buf("dec.ca.kv_raw_all", Nk, 2 * H * L, dtype=dt)
        nrm_all = self._buf("dec.ca.kv_n_all", Nk, 2 * H * L, dtype=dt)
        last = L if last is None else last                 # layers [first, last): a column slice of the all-layer buffers
        gains = [P[f"decoder.layers.{l}.cross_attn.{kv}_norm.weight"] for l in range(first, last) for kv in ("k", "v")]
        c0, c1 = 2 * H * first, 2 * H * last
        self._proj_headnorm(xkv, self._Wf(f"decoder.layers.{first}.cross_attn.w_k.weight", 2 * (last - first)), raw_all[:, c0:c1],
                            nrm_all[:, c0:c1], Sk, gains, 0, None, None)

    def _cross_kv_bwd_all(self, xkv, Nk, dt, d_xkv, wgrad=True, dgrad=True):
        """Weight gradient of all layers' K/V projections and the memory gradient: two GEMMs over the all-layer buffer."""
        a, H, L = self.arena, self.dims.hidden, self.dims.dec_layers
        draw_all = self._buf("dec.ca.dkv_raw_all", Nk, 2 * H * L, dtype=dt)
        if wgrad:
            with self._grouped_wgrads():                # (a group of one — the 128x64-tile, full-reduction launch — or a member of the
                self._wgrad(draw_all, xkv, a.fused(a.g, "decoder.layers.0.cross_attn.w_k.weight", 2 * L))   # caller's open group)
        if dgrad:
            self._dgrad(draw_all, self._Wf("decoder.layers.0.cross_attn.w_k.weight", 2 * L), d_xkv)

    def _attn_bwd(self, key, prefix, d_out, xq, xkv, B, Sq, Sk, rope, causal, key_mask, d_xq, d_xkv, d_xkv_beta,
                  site=0, p=0.0, dpr=0.0, layer=0):
        """Given d_out = dL/d(sub-layer output, pre-residual), accumulate parameter grads, write d_xq (dL/d xq) and,
        for cross-attention, d_xkv (+= when d_xkv_beta == 1)."""
        a, P, G, H, h = self.arena, self.arena.P, self.arena.G, self.dims.hidden, self.dims.heads
        Nq, Nk = B * Sq, B * Sk
        dt = xq.dtype
        i16 = _b16(xq)
        cos, sin = self._rope_tables(max(Sq, Sk)) if rope else (None, None)
        ctx, lse = self._buf(key + ".ctx", Nq, H, dtype=dt), self._buf(key + ".lse", B, h, Sq)
        # cross-attention: dctx / delta are read by the K/V branch on its own stream, so each layer keeps its own
        ck = "tmp" if xkv is None else key
        dctx, delta = self._buf(ck + ".dctx", Nq, H, dtype=dt), self._buf(ck + ".delta", B, h, Sq)
        # _tail_bwd (the fused LayerNorm backward before this call) already wrote the masked gradient of the projection
        # output and the column sums for w_o.bias
        d_out = self._buf("tmp.d_attn_proj" + ("" if xkv is None else ".x"), Nq, H, dtype=dt)
        self._wgrad(d_out, ctx, G[prefix + ".w_o.weight"], None)
        Wo = self._W(prefix + ".w_o.weight")
        # Delta = rowsum(dctx * ctx): from the epilogue of this GEMM when the pair launch below takes it as an input, else computed
        # by the dQ kernel from fragments it holds anyway (and read by the dK/dV kernel launched after it)
        pair = bool(self.attn_bwd_pair and self.fuse_headnorm_bwd and i16 and _b16(d_out) and _b16(Wo) and (xkv is None or d_xkv is not None)
                    and min(Sq, Sk) > self.attn_pair_min_seq and kk.load().kk_gemm_dgrad_delta_supported(Nq, H, H))
        if pair:
            kk.call("kk_gemm_dgrad_delta", Nq, H, H, d_out, d_out.stride(0), Wo, Wo.shape[1], dctx, dctx.stride(0), ctx, ctx.stride(0),
                    delta, Sq, h)
        else:
            self._dgrad(d_out, Wo, dctx)
        if xkv is None:
            raw, nrm = self._buf(key + ".qkv_raw", Nq, 3 * H, dtype=dt), self._buf(key + ".qkv_n", Nq, 3 * H, dtype=dt)
            dn, draw = self._buf("tmp.dqkv_n", Nq, 3 * H, dtype=dt), self._buf("tmp.dqkv_raw", Nq, 3 * H, dtype=dt)
            q_raw, k_raw, v_raw, q_n, k_n, v_n = raw, raw[:, H:], raw[:, 2 * H:], nrm, nrm[:, H:], nrm[:, 2 * H:]
            dq_n, dk_n, dv_n, dq_raw, dk_raw, dv_raw = dn, dn[:, H:], dn[:, 2 * H:], draw, draw[:, H:], draw[:, 2 * H:]
        else:
            q_raw, q_n = self._buf(key + ".q_raw", Nq, H, dtype=dt), self._buf(key + ".q_n", Nq, H, dtype=dt)
            kv_raw, kv_n = self._cross_kv(layer, Nk, dt)
            dq_n, dq_raw = self._buf("tmp.dq_n", Nq, H, dtype=dt), self._buf("tmp.dq_raw", Nq, H, dtype=dt)
            k_raw, v_raw, k_n, v_n = kv_raw, kv_raw[:, H:], kv_n, kv_n[:, H:]
        ld = lambda t: t.stride(0)
        gq, gk, gv = P[prefix + ".q_norm.weight"], P[prefix + ".k_norm.weight"], P[prefix + ".v_norm.weight"]
        dgq, dgk, dgv = G[prefix + ".q_norm.weight"], G[prefix + ".k_norm.weight"], G[prefix + ".v_norm.weight"]
        cz = 1 if causal else 0
        fuse = self.fuse_headnorm_bwd

        def hn_tables(tag, S, entries):
            """KkAttnHeadNorm descriptors of one backward launch (cached: every pointer is a persistent buffer) with their
            partial gain-gradient rows registered for the reduction at the end of the backward."""
            nb = kk.load().kk_attn_bwd_blocks(B, h, S)
            part = self._buf(f"{key}.hnpart.{tag}", len(entries), nb, 64)
            tk = ("hn", self._tmp_ns, key, tag, B, S, part.data_ptr()) + tuple(
                (r.data_ptr(), r.stride(0), g_.data_ptr(), c_.data_ptr() if c_ is not None else 0) for r, g_, _, c_, _ in entries)
            table = self._table(tk, lambda: kk.attn_headnorm([(r, g_, part[j], c_, s_) for j, (r, g_, _, c_, s_) in enumerate(entries)]))
            for j, (_, _, dg_, _, _) in enumerate(entries):
                self._reduce_lists[self._tmp_ns].append((part[j], dg_, None, nb, 64, 64))
            return table

        def warm_bwd():
            """the dQ half of the pair launch warms the weights of the dgrad GEMM(s) behind it (kk_attn_warm_next)"""
            if not (self.attn_warm & 12) or not (i16 and key.startswith("dec") and self.use_shadow):
                return
            w0 = self._Wf(prefix + ".w_q.weight", 3) if xkv is None else self._W(prefix + ".w_q.weight")
            w1 = None
            if self.attn_warm & 8:       # ... and of the output projection whose dgrad comes after the tail behind that (the self-attention's, from the cross-attention)
                base = prefix.rsplit(".", 1)[0]
                w1 = self._W(base + ".self_attn.w_o.weight") if xkv is not None else None
            kk.load().kk_attn_warm_next(w0.data_ptr(), w0.numel() * w0.element_size(), w1.data_ptr() if w1 is not None else None,
                                        w1.numel() * w1.element_size() if w1 is not None else 0)

        def bwd_one_call(*args):
            """kk_attn_bwd, or — where the library says the shape pays (kk_attn_bwd_two_pass: full attention from 1024 x 1024 scores per
            head up) — kk_attn_bwd_ws with this stream's dS workspace (2 bytes per score; "tmp.": private to the stream)."""
            if self.attn_two_pass and kk.load().kk_attn_bwd_two_pass(B, h, Sq, Sk, cz):
                need = kk.load().kk_attn_bwd_ws_bytes(B, h, Sq, Sk)
                kk.call("kk_attn_bwd_ws", *args, self._buf("tmp.attn_dS", need, dtype=torch.uint8), need)
                return
            keep = self._attn_keep(key, B, Sq, Sk, p, i16)          # (the buffer the forward of this sub-layer filled)
            warm_bwd()
            if keep is not None:
                kk.call("kk_attn_bwd_kb", *args, keep)
            else:
                kk.call("kk_attn_bwd", *args)

        if xkv is None:
            if pair:       # dQ | dK, dV in one launch, the head norms' backward as their epilogues
                bwd_one_call(q_n, k_n, v_n, dctx, lse, delta, dq_raw, dk_raw, dv_raw, B, h, Sq, Sk, ld(q_n), ld(k_n), ld(v_n), H,
                             ld(dq_raw), ld(dk_raw), ld(dv_raw), key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16,
                             hn_tables("q", Sq, [(q_raw, gq, dgq, cos, sin)]),
                             hn_tables("kv", Sk, [(k_raw, gk, dgk, cos, sin), (v_raw, gv, dgv, None, None)]))
            elif fuse:     # the head norms' backward is the epilogue of the two attention backward kernels
                kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, dctx, lse, delta, dq_raw, B, h, Sq, Sk, ld(q_n), ld(k_n), ld(v_n), H, ld(dq_raw),
                        key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16, ctx, H, hn_tables("q", Sq, [(q_raw, gq, dgq, cos, sin)]))
                kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, dctx, lse, delta, dk_raw, dv_raw, B, h, Sq, Sk, ld(q_n), ld(k_n), ld(v_n), H,
                        ld(dk_raw), ld(dv_raw), key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16,
                        hn_tables("kv", Sk, [(k_raw, gk, dgk, cos, sin), (v_raw, gv, dgv, None, None)]))
            else:
                kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, dctx, lse, delta, dq_n, B, h, Sq, Sk, ld(q_n), ld(k_n), ld(v_n), H, ld(dq_n),
                        key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16, ctx, H, None)
                kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, dctx, lse, delta, dk_n, dv_n, B, h, Sq, Sk, ld(q_n), ld(k_n), ld(v_n), H,
                        ld(dk_n), ld(dv_n), key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16, None)
                kk.call("kk_headnorm_rope_bwd", dn, 3 * H, raw, 3 * H, draw, 3 * H, Nq, h, Sq, 3, gq, gk, gv, dgq, dgk, dgv,
                        self._headnorm_partials(key, Nq, (dgq, dgk, dgv)), 3 if rope else 0, cos, sin, i16)
            self._wgrad(draw, xq, a.fused(a.g, prefix + ".w_q.weight", 3))
            self._dgrad(draw, self._Wf(prefix + ".w_q.weight", 3), d_xq)
            return
        if pair:
            dkv_raw, _ = self._cross_kv(layer, Nk, dt, "d")
            bwd_one_call(q_n, k_n, v_n, dctx, lse, delta, dq_raw, dkv_raw, dkv_raw[:, H:], B, h, Sq, Sk, ld(q_n), ld(k_n), ld(v_n), H,
                         ld(dq_raw), ld(dkv_raw), ld(dkv_raw), key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16,
                         hn_tables("q", Sq, [(q_raw, gq, dgq, None, None)]),
                         hn_tables("kv", Sk, [(k_raw, gk, dgk, None, None), (v_raw, gv, dgv, None, None)]))
        elif fuse:
            kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, dctx, lse, delta, dq_raw, B, h, Sq, Sk, ld(q_n), ld(k_n), ld(v_n), H, ld(dq_raw),
                    key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16, ctx, H,
                    hn_tables("q", Sq, [(q_raw, gq, dgq, None, None)]))      # also writes delta
            if d_xkv is not None:    # key/value branch; its two GEMMs run once for all layers (_cross_kv_bwd_all)
                dkv_raw, _ = self._cross_kv(layer, Nk, dt, "d")
                kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, dctx, lse, delta, dkv_raw, dkv_raw[:, H:], B, h, Sq, Sk, ld(q_n), ld(k_n),
                        ld(v_n), H, ld(dkv_raw), ld(dkv_raw), key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16,
                        hn_tables("kv", Sk, [(k_raw, gk, dgk, None, None), (v_raw, gv, dgv, None, None)]))
        else:
            kk.call("kk_attn_bwd_dq", q_n, k_n, v_n, dctx, lse, delta, dq_n, B, h, Sq, Sk, ld(q_n), ld(k_n), ld(v_n), H, ld(dq_n),
                    key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16, ctx, H, None)      # also writes delta
            if d_xkv is not None:
                dkv_raw, dkv_n = self._cross_kv(layer, Nk, dt, "d")
                kk.call("kk_attn_bwd_dkv", q_n, k_n, v_n, dctx, lse, delta, dkv_n, dkv_n[:, H:], B, h, Sq, Sk, ld(q_n), ld(k_n),
                        ld(v_n), H, ld(dkv_n), ld(dkv_n), key_mask, cz, 0.125, self.rng, site + 3, p, self.math, i16, None)
                kk.call("kk_headnorm_rope_bwd", dkv_n, ld(dkv_n), kv_raw, ld(kv_raw), dkv_raw, ld(dkv_raw), Nk, h, Sk, 2, gk, gv, None,
                        dgk, dgv, None, self._headnorm_partials(key + ".kv", Nk, (dgk, dgv)), 0, None, None, i16)
            kk.call("kk_headnorm_rope_bwd", dq_n, H, q_raw, H, dq_raw, H, Nq, h, Sq, 1, gq, None, None, dgq, None, None,
                    self._headnorm_partials(key + ".q", Nq, (dgq,)), 0, None, None, i16)
        self._wgrad(dq_raw, xq, G[prefix + ".w_q.weight"])
        self._dgrad(dq_raw, self._W(prefix + ".w_q.weight"), d_xq)

    # ------------------------------------------------------------------ text encoder: all layers in one launch
    def _encoder_stack_ok(self, B: int, Pn: int) -> bool:
        """The persistent launch carries batch item b on workgroup group b % 8: up to 8 items it is one pass (265-290 us against 545 us
        as 48 launches at 8 x 64 phonemes); a group walks further items one after the other, and from two passes on the 48 launches —
        whose kernels grow with the batch instead — are level or faster (dynamic batching, B up to 32: 9.50 against 9.69 ms per
        step on the configs[2] workload), so larger batches take the per-kernel sequence."""
        d = self.dims
        return bool(self.enc_fused and B <= self.enc_fused_max_batch and self.enc_dt == torch.bfloat16 and self.use_shadow and self.math == kk.KK_MATH_BF16 and
                    kk.load().kk_encoder_stack_supported(B, Pn, d.hidden, d.enc_ff, d.heads, d.enc_layers))

    def _encoder_stack_fwd(self, x0, B, Pn, text_mask, p_enc):
        """Forward of every encoder layer by kk_encoder_stack_fwd: same buffers, same masks as the per-kernel sequence
        (_attn_fwd / _ffn_fwd / _sublayer_tail), so the backward does not know the difference.  Returns (stream after the
        last layer, encoder_norm output)."""
        d, P, H, F, h, L = self.dims, self.arena.P, self.dims.hidden, self.dims.enc_ff, self.dims.heads, self.dims.enc_layers
        Ne, edt = B * Pn, self.enc_dt
        cos, sin = self._rope_tables(Pn)
        layers = []
        for i in range(L):
            pf, key = f"transformer_encoder_layers.{i}", f"enc{i}"
            nkey, npf, ndt = ((f"enc{i + 1}.ln1", f"transformer_encoder_layers.{i + 1}.norm1", edt) if i + 1 < L
                              else ("enc.norm", "encoder_norm", torch.float32))
            sa, ff = key + ".sa", key + ".ff"
            layers.append(dict(
                w_qkv=self._Wf(pf + ".self_attn.w_q.weight", 3), g_q=P[pf + ".self_attn.q_norm.weight"],
                g_k=P[pf + ".self_attn.k_norm.weight"], g_v=P[pf + ".self_attn.v_norm.weight"],
                w_o=self._W(pf + ".self_attn.w_o.weight"), b_o=P[pf + ".self_attn.w_o.bias"],
                ln2_g=P[pf + ".norm2.weight"], ln2_b=P[pf + ".norm2.bias"],
                w1=self._W(pf + ".ff.linear1.weight"), b1=P[pf + ".ff.linear1.bias"],
                w2=self._W(pf + ".ff.linear2.weight"), b2=P[pf + ".ff.linear2.bias"], ffn_gain=P[pf + ".ff.output_norm.weight"],
                next_g=P[npf + ".weight"], next_b=P[npf + ".bias"],
                y1=self._buf(key + ".ln1.y", Ne, H, dtype=edt),
                qkv_raw=self._buf(sa + ".qkv_raw", Ne, 3 * H, dtype=edt), qkv_n=self._buf(sa + ".qkv_n", Ne, 3 * H, dtype=edt),
                ctx=self._buf(sa + ".ctx", Ne, H, dtype=edt), lse=self._buf(sa + ".lse", B, h, Pn),
                proj=self._buf(sa + ".proj", Ne, H),
                x_in=x0 if i == 0 else self._buf(f"enc{i - 1}.xo", Ne, H), xm=self._buf(key + ".xm", Ne, H),
                y2=self._buf(key + ".ln2.y", Ne, H, dtype=edt), mean2=self._buf(key + ".ln2.mean", Ne), rstd2=self._buf(key + ".ln2.rstd", Ne),
                h1=self._buf(ff + ".h1", Ne, 2 * F, dtype=edt), g=self._buf(ff + ".g", Ne, F, dtype=edt),
                f2=self._buf(ff + ".f2", Ne, H, dtype=edt), rstd_f=self._buf(ff + ".rstd_f", Ne),
                xo=self._buf(key + ".xo", Ne, H), next_y=self._buf(nkey + ".y", Ne, H, dtype=ndt),
                next_mean=self._buf(nkey + ".mean", Ne), next_rstd=self._buf(nkey + ".rstd", Ne),
                next_y_bf16=1 if ndt == torch.bfloat16 else 0, site=1000 + 32 * i, p=float(p_enc), dpr=float(self._dpr(i, L))))
        sig = ("encstack", B, Pn, float(p_enc), self.train_dropout, self.enc_placement) + tuple(v.data_ptr() for v in layers[0].values() if hasattr(v, "data_ptr"))
        desc = self._table(sig, lambda: kk.enc_stack(B, Pn, H, F, h, text_mask, cos, sin, self.rng, self._enc_sync, layers, self.enc_placement,
                                                     self.enc_trace, self.enc_trace_wg))
        kk.call("kk_encoder_stack_fwd", desc)
        return self._buf(f"enc{L - 1}.xo", Ne, H), self._buf("enc.norm.y", Ne, H)

    # ------------------------------------------------------------------ GLU feed-forward sub-layer
    def _ffn_fwd(self, key, prefix, y, x_res, x_out, Fd, S=1, site=0, p=0.0, dpr=0.0, next_ln=None):
        P = self.arena.P
        N, H = y.shape
        dt, i16 = y.dtype, _b16(y)
        h1, g, f2 = (self._buf(key + ".h1", N, 2 * Fd, dtype=dt), self._buf(key + ".g", N, Fd, dtype=dt),
                     self._buf(key + ".f2", N, H, dtype=dt))
        W1 = self._W(prefix + ".linear1.weight")
        if i16 and _b16(W1) and H % 64 == 0 and self.fuse_glu_fwd:             # bf16 mode: the gate is the epilogue of the linear1 GEMM
            kk.call("kk_gemm_linear_glu", N, Fd, H, y, y.stride(0), W1, P[prefix + ".linear1.bias"], h1, g, Fd, self.rng, site + 4, p)
        else:
            self._linear(y, W1, P[prefix + ".linear1.bias"], h1)
            kk.call("kk_glu_fwd", h1, g, N, Fd, self.rng, site + 4, p, i16)
        self._linear(g, self._W(prefix + ".linear2.weight"), P[prefix + ".linear2.bias"], f2)
        # rmsnorm -> FFN dropout (:111) -> drop_path -> residual dropout (+ the next LayerNorm), one launch
        return self._sublayer_tail(f2, x_res, x_out, S, site, p, dpr, p, P[prefix + ".output_norm.weight"],
                                   self._buf(key + ".rstd_f", N), next_ln)

    def _ffn_bwd(self, key, prefix, d_out, y, d_y, Fd, S=1, site=0, p=0.0, dpr=0.0):
        P, G = self.arena.P, self.arena.G
        N, H = y.shape
        dt, i16 = y.dtype, _b16(y)
        h1, g, f2 = (self._buf(key + ".h1", N, 2 * Fd, dtype=dt), self._buf(key + ".g", N, Fd, dtype=dt),
                     self._buf(key + ".f2", N, H, dtype=dt))
        df2, dg, dh1 = (self._buf("tmp.df2", N, H, dtype=dt), self._buf("tmp.dg", N, Fd, dtype=dt),
                        self._buf("tmp.dh1", N, 2 * Fd, dtype=dt))
        # _tail_bwd already produced df2 (masks + RMSNorm backward) and the column sums for linear2.bias / output_norm
        self._wgrad(df2, g, G[prefix + ".linear2.weight"], None)
        W2 = self._W(prefix + ".linear2.weight")
        if i16 and _b16(W2) and H % 64 == 0:             # bf16 mode: the gate's backward is the epilogue of the linear2 dgrad
            nb = kk.load().kk_gemm_dgrad_glu_blocks(N)
            part = self._buf(key + ".glupart", nb, 2 * Fd)
            self._reduce_lists[self._tmp_ns].append((part, G[prefix + ".linear1.bias"], None, nb, 2 * Fd, 2 * Fd))
            kk.call("kk_gemm_dgrad_glu", N, Fd, H, df2, df2.stride(0), W2, h1, dh1, part, self.rng, site + 4, p)
            self._wgrad(dh1, y, G[prefix + ".linear1.weight"], None)
        else:
            self._dgrad(df2, W2, dg)
            kk.call("kk_glu_bwd", dg, h1, dh1, N, Fd, self.rng, site + 4, p, i16)
            self._wgrad(dh1, y, G[prefix + ".linear1.weight"], G[prefix + ".linear1.bias"])
        self._dgrad(dh1, self._W(prefix + ".linear1.weight"), d_y)

    # ------------------------------------------------------------------ variance predictor
    def _varpred_fwd(self, key, prefix, x, col1, B, L, mask, out, site=0, p=0.0):
        """x [B*L, H]; col1 = im2col3(x) (shared by pitch & energy predictors); out [B*L]."""
        P, Fv = self.arena.P, self.dims.var_filter
        rows, nch = B * L, -(-L // CHUNK)
        scratch = self._buf("tmp.gn_scratch", 2 * B * nch, dtype=torch.float64)
        inp_col, cin = col1, x.shape[1]
        for li in range(2):
            c, y = self._buf(f"{key}.c{li}", rows, Fv), self._buf(f"{key}.y{li}", rows, Fv)
            stats = self._buf(f"{key}.st{li}", B * nch, 2)
            self._linear(inp_col, self._Wconv(f"{prefix}.conv_layers.{li}.weight", Fv, 3 * cin), P[f"{prefix}.conv_layers.{li}.bias"], c)
            kk.call("kk_groupnorm_relu_fwd", c, P[f"{prefix}.norms.{li}.weight"], P[f"{prefix}.norms.{li}.bias"], y, stats,
                    scratch, B, L, Fv, CHUNK, self.rng, site + li, p)
            if li == 0:
                inp_col, cin = self._buf(f"{key}.col2", rows, 3 * Fv, dtype=col1.dtype), Fv
                kk.call("kk_im2col3_fwd", y, inp_col, B, L, Fv, CHUNK, _b16(inp_col))
        kk.call("kk_rowdot_fwd", y, P[f"{prefix}.linear.weight"], P[f"{prefix}.linear.bias"], mask, out, rows, Fv, L, CHUNK, 0)

    def _varpred_bwd(self, key, prefix, dout, x, col1, B, L, mask, dx, p=0.0):
        """Accumulate the predictor's parameter grads; write dx (dL/dx) when dx is not None."""
        P, G, Fv = self.arena.P, self.arena.G, self.dims.var_filter
        rows, nch, H = B * L, -(-L // CHUNK), x.shape[1]
        scratch = self._buf("tmp.gn_scratch", 2 * B * nch, dtype=torch.float64)
        # bf16 mode: the conv gradients' GEMM operand is stored as bf16 like every other dY of the step (DMA GEMM core)
        dc_dt = col1.dtype
        dy, dc = self._buf("tmp.vp_dy", rows, Fv), self._buf("tmp.vp_dc", rows, Fv, dtype=dc_dt)
        y1 = self._buf(f"{key}.y1", rows, Fv)
        kk.call("kk_rowdot_bwd", dout, y1, P[f"{prefix}.linear.weight"], mask, dy, G[f"{prefix}.linear.weight"],
                G[f"{prefix}.linear.bias"], rows, Fv, L, CHUNK, 0,
                self._rowdot_partials(key, rows, Fv, G[f"{prefix}.linear.weight"], G[f"{prefix}.linear.bias"]))
        for li in (1, 0):
            c, y, stats = self._buf(f"{key}.c{li}", rows, Fv), self._buf(f"{key}.y{li}", rows, Fv), self._buf(f"{key}.st{li}", B * nch, 2)
            cin = Fv if li == 1 else H
            col = self._buf(f"{key}.col2", rows, 3 * Fv, dtype=col1.dtype) if li == 1 else col1
            kk.call("kk_groupnorm_relu_bwd", dy, c, y, P[f"{prefix}.norms.{li}.weight"], stats, dc, G[f"{prefix}.norms.{li}.weight"],
                    G[f"{prefix}.norms.{li}.bias"], scratch, B, L, Fv, CHUNK, p, _b16(dc))
            W, dW = self._Wconv(f"{prefix}.conv_layers.{li}.weight", Fv, 3 * cin), G[f"{prefix}.conv_layers.{li}.weight"].view(Fv, 3 * cin)
            self._wgrad(dc, col, dW, G[f"{prefix}.conv_layers.{li}.bias"])
            if li == 1 or dx is not None:
                dcol = self._buf(f"tmp.vp_dcol{li}", rows, 3 * cin, dtype=col1.dtype)
                self._dgrad(dc, W, dcol)
                kk.call("kk_im2col3_bwd", dcol, dy if li == 1 else dx, B, L, cin, CHUNK, _b16(dcol))

    # ------------------------------------------------------------------ forward + losses + backward
    def forward_backward(self, batch: Dict[str, torch.Tensor], loss_scale: float = 1.0, adaptive: bool = False,
                         backward: bool = True, zero_grads: bool = False, expanded_len: Optional[int] = None
                         ) -> Dict[str, torch.Tensor]:
        """One micro-batch: forward, the 6 losses, and (optionally) the full backward into the gradient arena
        (which accumulates; zero_grads=True clears it first, overlapped with the forward).  `batch` follows the
        reference collate contract (data/dataset.py:871-921), tensors on the device.  Returns device tensors (no sync):
        losses[6] = (total, mel, dur, stop, pitch, energy) and outputs.

        expanded_len = max_b sum(phoneme_durations[b]), the frame count T' the durations expand to, as a host integer
        (the caller has the durations on the host before the copy; nothing in a step synchronises).  None means T' = T,
        which holds for the reference's datasets unless an utterance was clipped (data/dataset.py:769-776).  T' > T:
        the pitch / energy predictors run on all T' frames and their outputs are [B, T'] like the reference's, the
        decoder memory is the first T frames (model/model.py:607-613), the losses read the first T columns
        (losses.py:111,137).  T' < T: the reference fails in the pitch loss with a size-mismatch RuntimeError
        (losses.py:126: [B, T'] predictions against [B, T] targets) — the same error is raised here, before any launch."""
        return self._fb(batch, loss_scale, adaptive, backward, zero_grads, expanded_len)

    def _fb(self, batch, loss_scale, adaptive, backward, zero_grads=False, expanded_len=None):
        """The launch sequence of forward_backward (also what train_step_graphed captures)."""
        self._defer_wgrads = None                     # (a list only while layer 0's grouped launch is being collected: ADVICE r5)
        self._ss_step = None
        try:
            with self._acc_guard():
                return self._fb_launches(batch, loss_scale, adaptive, backward, zero_grads, expanded_len)
        finally:
            self._defer_wgrads = None
            self._ss_step = None
            self._wgrad_queue.clear()

    def _fb_launches(self, batch, loss_scale, adaptive, backward, zero_grads=False, expanded_len=None):
        d, a, P, G = self.dims, self.arena, self.arena.P, self.arena.G
        self._grads_fresh = bool(zero_grads) or self._first_micro
        self._ow_seen = set() if (self._grads_fresh and backward) else None
        self._ow_expect = None
        H, M, Fv = d.hidden, d.mel, d.var_filter
        ids, mel, dur = batch["phoneme_indices"], batch["mel_specs"], batch["phoneme_durations"]
        stress = batch.get("stress_indices")
        B, Pn = ids.shape
        T = mel.shape[1]
        Ne, Nd = B * Pn, B * T
        edt, ddt = self.enc_dt, self.dec_dt               # activation storage of the encoder / decoder stacks
        pe = P["positional_encoding.pe"].view(d.max_len, H)
        Tp = T if expanded_len is None else max(int(expanded_len), 3)     # variance_predictor.py:358-360 (at least 3 frames)
        if Tp < T:
            raise RuntimeError(f"The size of tensor a ({Tp}) must match the size of tensor b ({T}) at non-singleton dimension 1 "
                               f"(durations expand to {Tp} frames, the batch has {T} mel frames: losses.py:126)")
        if max(T, Tp) > d.max_len or Pn > d.max_len:
            raise ValueError(f"sequence longer than the positional table ({d.max_len})")
        Np = B * Tp

        # ---- decoder head: mel input projection + layer-0 self-attention need no encoder output (model.py:519-531) ----
        p_din = self._p(self.hp.decoder_input_dropout)
        shifted = self._buf("dec.shifted", Nd, M)

        def self_attn(i, y_in, n1_in):
            key, pf, st = f"dec{i}", f"decoder.layers.{i}", 2000 + 32 * i
            ya_ = self._buf(key + ".xa", Nd, H)
            n2_ = self._attn_fwd(key + ".sa", pf + ".self_attn", n1_in, None, B, T, T, True, True, None, y_in, ya_, st, p_dec,
                                 self._dpr(i, d.dec_layers), next_ln=(key + ".ln2", pf + ".norm2", ddt))
            return ya_, (n2_ if n2_ is not None else self._ln_fwd(key + ".ln2", ya_, pf + ".norm2", ddt))

        def decoder_head():
            kk.call("kk_shift_right", mel, shifted, B, T, M)
            y0 = self._buf("dec.x0", Nd, H)
            if self.train_dropout and (p_din > 0.0 or pe_drop > 0.0):    # dropout(proj) + pe, then the PE module's dropout
                lin, t1 = self._buf("tmp.dec_lin", Nd, H), self._buf("tmp.dec_t1", Nd, H)
                self._linear(shifted, self._W("mel_projection_in.weight"), P["mel_projection_in.bias"], lin)
                kk.call("kk_dropout_fwd", lin, pe, T, t1, Nd, H, T, self.rng, 30, p_din, 0, 0.0, 0, 0.0)
                kk.call("kk_dropout_fwd", t1, None, 0, y0, Nd, H, T, self.rng, 31, pe_drop, 0, 0.0, 0, 0.0)
            else:
                self._linear(shifted, self._W("mel_projection_in.weight"), P["mel_projection_in.bias"], y0, res=pe, res_mod=T)
            n10 = self._ln_fwd("dec0.ln1", y0, "decoder.layers.0.norm1", ddt)
            # The 200 MB gradient zero-fill goes HERE: the launches above are in flight before the persistent encoder
            # kernel (which takes every CU's LDS) starts; the fill needs no LDS and runs beside it, while the GEMMs and
            # the attention below cannot and follow the encoder — with the fill in front of the whole head, none of the
            # head ran early and it ended 46 us after the cross-attention K/V GEMM, on the critical path.
            if zero_grads and self.zero_late:
                self._zero_grad_step()
            self._mark("kv: zero_grad done")
            if gen_keep:
                # the keep bits of every decoder attention launch of this step, HERE: this stream idles until the persistent encoder
                # launch (which holds every CU's LDS) has ended, and the generator needs no LDS — it runs beside the encoder
                self._keep_gen_launch(B, T, p_dec, 0, True, layers=gen_layers)
                self._mark("kv: keep bits generated")
            ya0, n20 = self_attn(0, y0, n10)
            return y0, ya0, n20

        # ---- encoder (model.py:375-388) ----
        text_mask = self._buf("text_mask", B, Pn, dtype=torch.uint8)
        if not self.fuse_enc_prologue:
            kk.call("kk_ids_eq_zero", ids, text_mask, Ne)
        x = self._buf("enc.x0", Ne, H)
        hp = self.hp
        if self.train_dropout:
            self.rng.add_(1)                              # fresh masks every micro-batch (captured in the hipGraph)
        pe_drop, p_enc, p_dec, p_var = self._p(hp.encoder_dropout), self._p(hp.encoder_dropout), self._p(hp.decoder_dropout), self._p(hp.variance_dropout)
        self._mark("step.start")
        stack = self._encoder_stack_ok(B, Pn)
        self._keep_ready = set()
        # beside the persistent encoder forward, where that pays (measured: -0.4 % at 8 x 512, +1.6 % at 8 x 1024, where the generator
        # outlasts the encoder and delays the decoder head; profiles/r06_keep_bits_gen_ab.txt)
        gen_keep = bool(self.attn_keep_gen and self.attn_keep_bits and self.train_dropout and p_dec > 0.0 and ddt == torch.bfloat16 and
                        self.overlap and stack and self.dec_head_aside)
        gen_layers = 0
        if gen_keep:
            units = B * self.dims.heads * ((T + 31) // 32) ** 2 * 1.5          # a layer: the causal self-attention (half) + the cross-attention
            gen_layers = min(self.dims.dec_layers, int(self.attn_keep_gen_rate * (160.0 + 2.0 * Pn) / units))
            gen_keep = gen_layers > 0
        with self._on_stream(self._kv, "kv.", self.dec_head_aside):     # beside the encoder; joined before the first cross-attention
            if zero_grads and not self.zero_late:
                self._zero_grad_step()
            kk.call("kk_max_i64", dur, Ne, self.max_dur)  # (read by the losses only: not in front of the encoder)
            dec_head = decoder_head()                     # (includes the gradient zero-fill, see there)
            self._mark("kv: decoder head done")
        y1 = None                                         # LayerNorm outputs come from the previous sub-layer's fused tail
        if self.fuse_enc_prologue:                        # key mask + embedding + layer 0's pre-norm: ONE launch in front of the encoder
            y1, m1, r1 = self._buf("enc0.ln1.y", Ne, H, dtype=edt), self._buf("enc0.ln1.mean", Ne), self._buf("enc0.ln1.rstd", Ne)
            kk.call("kk_embed_ln_fwd", ids, stress, P["text_embedding.weight"], P["stress_embedding.weight"] if stress is not None else None,
                    pe, x, B, Pn, H, float(H ** 0.5), self.rng, 1, pe_drop, text_mask, P["transformer_encoder_layers.0.norm1.weight"],
                    P["transformer_encoder_layers.0.norm1.bias"], y1, _b16(y1), m1, r1)
        else:
            kk.call("kk_embed_fwd", ids, stress, P["text_embedding.weight"], P["stress_embedding.weight"] if stress is not None else None,
                    pe, x, B, Pn, H, float(H ** 0.5), self.rng, 1, pe_drop)
        if stack:                                         # all layers in ONE persistent launch (csrc/kk_encstack.hip)
            if y1 is None:
                y1 = self._ln_fwd("enc0.ln1", x, "transformer_encoder_layers.0.norm1", edt)
            x, y1 = self._encoder_stack_fwd(x, B, Pn, text_mask, p_enc)
            self._mark(f"enc{d.enc_layers - 1} fwd done")
        for i in range(0 if not stack else d.enc_layers, d.enc_layers):      # when dropout is on, from _ln_fwd otherwise
            pf, key, st = f"transformer_encoder_layers.{i}", f"enc{i}", 1000 + 32 * i
            dpr = self._dpr(i, d.enc_layers)
            if y1 is None:
                y1 = self._ln_fwd(key + ".ln1", x, pf + ".norm1", edt)
            xm = self._buf(key + ".xm", Ne, H)
            y2 = self._attn_fwd(key + ".sa", pf + ".self_attn", y1, None, B, Pn, Pn, True, False, text_mask, x, xm, st, p_enc, dpr,
                                next_ln=(key + ".ln2", pf + ".norm2", edt))
            if y2 is None:
                y2 = self._ln_fwd(key + ".ln2", xm, pf + ".norm2", edt)
            xo = self._buf(key + ".xo", Ne, H)
            nxt = ((f"enc{i + 1}.ln1", f"transformer_encoder_layers.{i + 1}.norm1", edt) if i + 1 < d.enc_layers
                   else ("enc.norm", "encoder_norm", torch.float32))
            y1 = self._ffn_fwd(key + ".ff", pf + ".ff", y2, xm, xo, d.enc_ff, Pn, st + 8, p_enc, dpr, next_ln=nxt)
            x = xo
            self._mark(f"enc{i} fwd done")
        enc_last = x
        enc = y1 if y1 is not None else self._ln_fwd("enc.norm", enc_last, "encoder_norm")

        # ---- variance adaptor (variance_predictor.py:338-439) ----
        dur_pred = self._buf("out.log_dur", B, Pn)
        col_e = self._buf("vp.col_enc", Ne, 3 * H, dtype=edt)      # (both conv unfolds are the predictors' inputs only: side stream)
        # (teacher forcing: the regulator and the embeddings below use the batch's durations / pitch / energy, so the three
        #  predictors feed only the losses and run on the side stream — see predictors())
        idx, lens, tot = (self._buf("lr.idx", B, T, dtype=torch.int64), self._buf("lr.lens", B, dtype=torch.int64),
                          self._buf("lr.total", B, dtype=torch.int64))
        kk.call("kk_length_regulate_index", dur, idx, lens, tot, B, Pn, T)
        xf = self._buf("va.xf", Nd, H)
        memory, fmask = self._buf("va.memory", Nd, H, dtype=ddt), self._buf("va.fmask", B, T, dtype=torch.uint8)
        pidx, eidx = self._buf("va.pidx", B, T, dtype=torch.int32), self._buf("va.eidx", B, T, dtype=torch.int32)
        spec_aug = self.train_dropout and hp.use_spec_augment and self.spec_augment_active
        if self.fuse_memory_fwd:     # gather + embedding adds + SpecAugment: ONE launch between the encoder and the cross-attention K/V GEMM
            kk.call("kk_regulate_embed_fwd", enc, idx, batch["pitches"], batch["energies"], P[f"{VA}.pitch_bins"], P[f"{VA}.energy_bins"],
                    P[f"{VA}.pitch_embedding.weight"], P[f"{VA}.energy_embedding.weight"], lens, xf, memory, pidx, eidx, fmask, B, Pn, T, H,
                    d.var_bins, _b16(memory), self.rng if spec_aug else None, 20, hp.spec_augment_time_mask_max,
                    hp.spec_augment_freq_mask_max, hp.spec_augment_num_time_masks, hp.spec_augment_num_freq_masks)
        else:
            kk.call("kk_length_regulate_gather", enc, idx, xf, B, Pn, T, H)       # detached by construction
            kk.call("kk_bucket_embed_add_fwd", xf, batch["pitches"], batch["energies"], P[f"{VA}.pitch_bins"], P[f"{VA}.energy_bins"],
                    P[f"{VA}.pitch_embedding.weight"], P[f"{VA}.energy_embedding.weight"], lens, memory, pidx, eidx, fmask, B, T, H,
                    d.var_bins, _b16(memory))
        pitch_pred, energy_pred = self._buf("out.pitch", B, Tp), self._buf("out.energy", B, Tp)
        col_f = self._buf("vp.col_frames", Np, 3 * H, dtype=ddt)
        if Tp != T:      # the predictors' own T'-frame view of the expansion; the losses read the first T columns
            xf_p, fmask_p = self._buf("va.xf_p", Np, H), self._buf("va.fmask_p", B, Tp, dtype=torch.uint8)
            pitch_l, energy_l = self._buf("out.pitch_T", B, T), self._buf("out.energy_T", B, T)
        else:
            xf_p, fmask_p, pitch_l, energy_l = xf, fmask, pitch_pred, energy_pred
        if spec_aug and not self.fuse_memory_fwd:         # on the cross-attention memory only (model.py:636-639)
            kk.call("kk_specaug", memory, B, T, H, self.rng, 20, hp.spec_augment_time_mask_max, hp.spec_augment_freq_mask_max,
                    hp.spec_augment_num_time_masks, hp.spec_augment_num_freq_masks, _b16(memory))

        # ---- decoder (model.py:519-531; transformers.py:543-583,660) ----
        self._mark("memory ready")
        self._cross_kv_fwd_all(memory, Nd, T, ddt)        # every layer's cross-attention K/V in one GEMM
        self._mark("cross K/V fwd done")
        # Forked only here, after the K/V GEMM: started earlier (right after im2col3) the predictors' fp32 GEMMs compete
        # with the critical path into the decoder; 2.8 % of the step (729K -> 749K frames/s).
        with self._on_side_stream():                      # joined before the losses
            self._mark("side: predictors fwd start")
            if backward and self.embed_bwd_sorted and d.var_bins <= 1024:
                n_items = kk.load().kk_bucket_sort_items(Nd, d.var_bins)
                be_order, be_items = self._buf("va.be_order", 2, Nd, dtype=torch.int32), self._buf("va.be_items", 2, n_items, 4, dtype=torch.int32)
                kk.call("kk_bucket_sort", pidx, eidx, fmask, Nd, d.var_bins, be_order, be_items)
            kk.call("kk_im2col3_fwd", enc, col_e, B, Pn, H, CHUNK, _b16(col_e))
            if Tp != T:
                idx_p, lens_p = self._buf("lr.idx_p", B, Tp, dtype=torch.int64), self._buf("lr.lens_p", B, dtype=torch.int64)
                kk.call("kk_length_regulate_index", dur, idx_p, lens_p, self._buf("lr.total_p", B, dtype=torch.int64), B, Pn, Tp)
                kk.call("kk_length_regulate_gather", enc, idx_p, xf_p, B, Pn, Tp, H)
                kk.call("kk_frame_mask", lens_p, fmask_p, B, Tp)
            kk.call("kk_im2col3_fwd", xf_p, col_f, B, Tp, H, CHUNK, _b16(col_f))
            self._varpred_fwd("vp.dur", f"{VA}.duration_predictor", enc, col_e, B, Pn, text_mask, dur_pred, 10, p_var)
            self._varpred_fwd("vp.pitch", f"{VA}.pitch_predictor", xf_p, col_f, B, Tp, fmask_p, pitch_pred, 12, p_var)
            self._varpred_fwd("vp.energy", f"{VA}.energy_predictor", xf_p, col_f, B, Tp, fmask_p, energy_pred, 14, p_var)
            if Tp != T:
                kk.call("kk_pad2d_f32", pitch_pred, Tp, Tp, pitch_l, T, T, B)
                kk.call("kk_pad2d_f32", energy_pred, Tp, Tp, energy_l, T, T, B)
            self._mark("side: predictors fwd done")
        n1 = None
        for i in range(d.dec_layers):
            pf, key, st = f"decoder.layers.{i}", f"dec{i}", 2000 + 32 * i
            dpr = self._dpr(i, d.dec_layers)
            if i == 0:                                    # input projection + layer-0 self-attention ran beside the encoder
                self._join(self._kv)
                y, ya, n2 = dec_head
            else:
                if n1 is None:
                    n1 = self._ln_fwd(key + ".ln1", y, pf + ".norm1", ddt)
                ya, n2 = self_attn(i, y, n1)
            yc = self._buf(key + ".xc", Nd, H)
            n3 = self._attn_fwd(key + ".ca", pf + ".cross_attn", n2, memory, B, T, T, False, False, fmask, ya, yc, st + 8, p_dec, dpr,
                                next_ln=(key + ".ln3", pf + ".norm3", ddt), layer=i)
            if n3 is None:
                n3 = self._ln_fwd(key + ".ln3", yc, pf + ".norm3", ddt)
            yo = self._buf(key + ".xo", Nd, H)
            nxt = ((f"dec{i + 1}.ln1", f"decoder.layers.{i + 1}.norm1", ddt) if i + 1 < d.dec_layers
                   else ("dec.norm", "decoder.norm", ddt))
            n1 = self._ffn_fwd(key + ".ff", pf + ".ff", n3, yc, yo, d.dec_ff, T, st + 16, p_dec, dpr, next_ln=nxt)
            y = yo
            self._mark(f"dec{i} fwd done")
        dec_last = y
        dec_out = n1 if n1 is not None else self._ln_fwd("dec.norm", dec_last, "decoder.norm", ddt)
        mel_pred, stop = self._buf("out.mel", B, T, M), self._buf("out.stop", B, T)
        self._linear(dec_out, self._W("mel_projection_out.weight"), P["mel_projection_out.bias"], mel_pred.view(Nd, M))
        kk.call("kk_rowdot_fwd", dec_out, P["stop_token_predictor.weight"], P["stop_token_predictor.bias"], None, stop, Nd, H, T, 0,
                _b16(dec_out))

        # ---- losses (losses.py) ----
        self._join_side()
        lcfg = kk.KkLossCfg(hp.duration_loss_weight, hp.stop_token_loss_weight, hp.pitch_loss_weight, hp.energy_loss_weight,
                            hp.duration_huber_delta, hp.pitch_huber_delta, hp.energy_huber_delta, hp.stop_token_pos_weight,
                            float(loss_scale), 1 if adaptive else 0)
        largs = (mel_pred, mel, dur_pred, dur, stop, batch["stop_token_targets"], pitch_l, batch["pitches"], energy_l,
                 batch["energies"], batch["mel_lengths"], batch["phoneme_lengths"], B, T, Pn, M, lcfg)
        # per-micro-batch finite-output / finite-loss guard (trainer.py:3233-3296), device side: a flagged micro-batch
        # back-propagates nothing and its accumulation cycle takes no optimizer step; forward-only calls (validation) do
        # not touch the training cycle's flag
        guard = self.opt_state[kk.OS["MICRO_BAD"]:] if backward else None
        # (loss_acc is handed round zero: allocated zero, cleared by its last reader — no zero-fill launch on the critical chain)
        sc = 1 if self.self_cleaning_acc else 0
        kk.call("kk_losses_fwd", *largs, self.max_dur, self.loss_acc, self.losses, self.loss_coef,
                guard if self.loss_sync is None else None, sc * (1 if self.loss_sync is not None else 3))
        if self.loss_sync is not None:                    # global normalisers for ragged shards (one more collective of the step)
            self.loss_sync.loss_sync(self.loss_acc, self.max_dur)
            kk.call("kk_losses_finalize", self.loss_acc, lcfg, self.max_dur, canonical_mel_length(self.global_mel_length, T), self.losses,
                    self.loss_coef, guard, sc)
        out = {"losses": self.losses, "mel": mel_pred, "log_dur": dur_pred, "stop": stop, "pitch": pitch_pred,
               "energy": energy_pred, "lr_idx": idx, "lr_lens": lens, "memory": memory.view(B, T, H)}
        if not backward:
            return out

        # =========================== backward ===========================
        self._ss_ready = None
        if self.grad_norm_from_wgrads and self.dp_comm is None and not self._external_sync and self.group_wgrads:
            self._ss_step = dict(rec=kk.C.c_void_p(self.sumsq_ws.data_ptr() + int(kk.load().kk_seg_sumsq_rec_offset())), count=kk.C.c_int32(0),
                                 cap=int(kk.load().kk_seg_sumsq_rec_capacity()), covered=set())
        for ns in self._reduce_lists:
            self._reduce_lists[ns] = []
        if self.dp_comm is not None and self._exchange_now:
            self.dp_comm.begin_step()
            self._comm_events = {}
        dmel, ddur = self._buf("g.mel", B, T, M), self._buf("g.dur", B, Pn)
        dstop, dpitch, denergy = self._buf("g.stop", B, T), self._buf("g.pitch", B, T), self._buf("g.energy", B, T)
        kk.call("kk_losses_bwd", *largs, self.loss_coef, dmel, ddur, dstop, dpitch, denergy)
        self._mark("losses + loss gradients done")
        fork = self._fork_point()

        def heads_and_frame_predictors():
            # the output heads' weight gradients have no consumer on the decoder chain (the stop head's input is
            # detached, model.py:561-562): 50 us off the critical path
            kk.call("kk_rowdot_bwd", dstop, dec_out, P["stop_token_predictor.weight"], None, None, G["stop_token_predictor.weight"],
                    G["stop_token_predictor.bias"], Nd, H, T, 0, _b16(dec_out),
                    self._rowdot_partials("stop", Nd, H, G["stop_token_predictor.weight"], G["stop_token_predictor.bias"]))
            self._wgrad(dmel.view(Nd, M), dec_out, G["mel_projection_out.weight"], G["mel_projection_out.bias"])
            dpitch_p, denergy_p = dpitch, denergy
            if Tp != T:                               # frames past T carry no loss: zero gradient there
                dpitch_p, denergy_p = self._buf("g.pitch_p", B, Tp), self._buf("g.energy_p", B, Tp)
                kk.call("kk_pad2d_f32", dpitch, T, T, dpitch_p, Tp, Tp, B)
                kk.call("kk_pad2d_f32", denergy, T, T, denergy_p, Tp, Tp, B)
            self._varpred_bwd("vp.pitch", f"{VA}.pitch_predictor", dpitch_p, xf_p, col_f, B, Tp, fmask_p, None, p_var)
            self._varpred_bwd("vp.energy", f"{VA}.energy_predictor", denergy_p, xf_p, col_f, B, Tp, fmask_p, None, p_var)

        def side_backward(after):                         # independent of the decoder backward (disjoint gradient segments)
            with self._on_side_stream(after=after):
                self._mark("side: backward start")
                heads_and_frame_predictors()
                d_enc = self._buf("g.enc_out", Ne, H)
                self._varpred_bwd("vp.dur", f"{VA}.duration_predictor", ddur, enc, col_e, B, Pn, text_mask, d_enc, p_var)
                self._mark("side: predictors bwd done")
                dx = self._buf("g.enc_stream", Ne, H)
                # every LayerNorm's backward is fused with the head of the backward of the sub-layer that produced its input
                ehead = lambda kind, i: (kind, f"enc{i}.ff", f"transformer_encoder_layers.{i}" + (".ff" if kind == "ffn" else ".self_attn"),
                                         Pn, 1000 + 32 * i + (8 if kind == "ffn" else 0), p_enc, self._dpr(i, d.enc_layers), edt)
                self._tail_bwd("enc.norm", d_enc, enc_last, "encoder_norm", dx, False, ehead("ffn", d.enc_layers - 1))
                dne = self._buf("tmp.dne", Ne, H, dtype=edt)
                for i in reversed(range(d.enc_layers)):
                    pf, key, st = f"transformer_encoder_layers.{i}", f"enc{i}", 1000 + 32 * i
                    dpr = self._dpr(i, d.enc_layers)
                    x_in = self._buf(f"enc{i - 1}.xo", Ne, H) if i > 0 else self._buf("enc.x0", Ne, H)
                    xm = self._buf(key + ".xm", Ne, H)
                    y1, y2 = self._buf(key + ".ln1.y", Ne, H, dtype=edt), self._buf(key + ".ln2.y", Ne, H, dtype=edt)
                    with self._grouped_wgrads():
                        self._ffn_bwd(key + ".ff", pf + ".ff", dx, y2, dne, d.enc_ff, Pn, st + 8, p_enc, dpr)
                        self._tail_bwd(key + ".ln2", dne, xm, pf + ".norm2", dx, True, ehead("attn", i))
                        self._attn_bwd(key + ".sa", pf + ".self_attn", dx, y1, None, B, Pn, Pn, True, False, text_mask, dne, None, 0.0,
                                       st, p_enc, dpr)
                    self._comm_bucket(f"enc{i}")            # (the grouped weight-gradient launch is behind us)
                    if i > 0:
                        self._tail_bwd(key + ".ln1", dne, x_in, pf + ".norm1", dx, True, ehead("ffn", i - 1))
                    else:
                        self._ln_bwd(key + ".ln1", dne, x_in, pf + ".norm1", dx, accumulate=True)
                    self._mark(f"side: enc{i} bwd done")
                kk.call("kk_embed_bwd", ids, stress, dx, G["text_embedding.weight"],
                        G["stress_embedding.weight"] if stress is not None else None, B, Pn, H, float(H ** 0.5), self.rng, 1, pe_drop)
                self._mark("side: backward done")
                if tail_mode in (3, 4):                   # the memory tail behind the encoder's backward (no third branch)
                    torch.cuda.current_stream().wait_event(tail_fork)
                    memory_tail()
                    if deferred0:
                        torch.cuda.current_stream().wait_event(wgrad0_fork)
                        for q in deferred0:
                            self._issue_wgrad_group(q)

        # The decoder backward is the critical path, so it is captured first and the predictors' + encoder's backward
        # after it, forked from the point right after the loss gradients (see _on_stream).
        # heads: only the mel projection's input gradient continues down the decoder (the two heads' weight gradients are
        # on the side branch, see side_backward)
        d_dec_out = self._buf("tmp.d_dec_out", Nd, H, dtype=ddt)
        self._dgrad(dmel.view(Nd, M), self._W("mel_projection_out.weight"), d_dec_out)
        dy = self._buf("g.dec_stream", Nd, H)          # gradient of the decoder residual stream, updated in place
        def dhead(kind, i):                            # (kind, ffn buffer key, parameter prefix, S, site, p, drop-path, dtype)
            sub = {"ffn": (".ff", 16), "ca": (".cross_attn", 8), "sa": (".self_attn", 0)}[kind]
            return ("ffn" if kind == "ffn" else "attn", f"dec{i}.ff", f"decoder.layers.{i}" + sub[0], T, 2000 + 32 * i + sub[1],
                    p_dec, self._dpr(i, d.dec_layers), ddt)
        self._tail_bwd("dec.norm", d_dec_out, dec_last, "decoder.norm", dy, False, dhead("ffn", d.dec_layers - 1))
        dmem = self._buf("g.memory", Nd, H)
        tail_mode = (4 if Nd <= 4096 else 3) if self.tail_aside < 0 else self.tail_aside
        tail_mode = tail_mode if self.overlap else 0
        deferred0 = []
        dn = self._buf("tmp.dn", Nd, H, dtype=ddt)
        for i in reversed(range(d.dec_layers)):
            pf, key, st = f"decoder.layers.{i}", f"dec{i}", 2000 + 32 * i
            dpr = self._dpr(i, d.dec_layers)
            x_in = self._buf(f"dec{i - 1}.xo", Nd, H) if i > 0 else self._buf("dec.x0", Nd, H)
            ya, yc = self._buf(key + ".xa", Nd, H), self._buf(key + ".xc", Nd, H)
            n1, n2, n3 = (self._buf(f"{key}.ln{j}.y", Nd, H, dtype=ddt) for j in (1, 2, 3))
            # layer 0's grouped weight gradients: nothing on the main chain overwrites their operands any more (the layer-norm backward and
            # the input projection below use their own buffers), so the launch can wait for the side stream's idle end (wgrad0_aside)
            defer0 = i == 0 and tail_mode in (3, 4) and self.group_wgrads and self.dp_comm is None and \
                (self.wgrad0_aside > 0 or (self.wgrad0_aside < 0 and tail_mode == 3))
            if defer0:
                self._defer_wgrads = deferred0 = []           # (reset by _fb whatever happens in between)
            with self._grouped_wgrads():
                self._ffn_bwd(key + ".ff", pf + ".ff", dy, n3, dn, d.dec_ff, T, st + 16, p_dec, dpr)
                self._tail_bwd(key + ".ln3", dn, yc, pf + ".norm3", dy, True, dhead("ca", i))
                self._attn_bwd(key + ".ca", pf + ".cross_attn", dy, n2, memory, B, T, T, False, False, fmask, dn, dmem,
                               0.0, st + 8, p_dec, dpr, layer=i)
                if i == 0 and tail_mode in (1, 3, 4):  # every layer's cross-attention dK / dV is written: the memory tail may start
                    tail_fork = self._fork_point()
                    if tail_mode == 4:            # (4: the K/V weight gradient as one more member of layer 0's grouped launch)
                        self._cross_kv_bwd_all(memory, Nd, ddt, dmem, dgrad=False)
                self._tail_bwd(key + ".ln2", dn, ya, pf + ".norm2", dy, True, dhead("sa", i))
                self._attn_bwd(key + ".sa", pf + ".self_attn", dy, n1, None, B, T, T, True, True, None, dn, None, 0.0, st, p_dec, dpr)
            if defer0:
                self._defer_wgrads = None
                wgrad0_fork = self._fork_point()
            self._comm_bucket(f"dec{i}")                    # the layer's weight matrices are final: exchange beside the rest
            if i > 0:
                self._tail_bwd(key + ".ln1", dn, x_in, pf + ".norm1", dy, True, dhead("ffn", i - 1))
            else:
                self._ln_bwd(key + ".ln1", dn, x_in, pf + ".norm1", dy, accumulate=True)
            self._mark(f"dec{i} bwd done")
        def input_projection_bwd():                    # (the PE add and the shift are parameter-free; mel is data)
            if self.train_dropout and (p_din > 0.0 or pe_drop > 0.0):
                t1, dlin = self._buf("tmp.d_dec_t1", Nd, H), self._buf("tmp.d_dec_lin", Nd, H)
                kk.call("kk_dropout_bwd", dy, t1, Nd, H, T, self.rng, 31, pe_drop, 0, 0.0, 0, 0.0, 0)
                kk.call("kk_dropout_bwd", t1, dlin, Nd, H, T, self.rng, 30, p_din, 0, 0.0, 0, 0.0, 0)
                self._wgrad(dlin, shifted, G["mel_projection_in.weight"], G["mel_projection_in.bias"])
            else:
                self._wgrad(dy, shifted, G["mel_projection_in.weight"], G["mel_projection_in.bias"])
            self._mark("decoder input projection bwd done")

        def memory_tail():
            # variance adaptor: memory gradient feeds only the two embedding tables (xf is detached, lengths.py:30)
            self._cross_kv_bwd_all(memory, Nd, ddt, dmem, wgrad=tail_mode != 4)   # K/V weight gradients and the memory gradient, all layers at once
            self._mark("cross K/V bwd (all layers) done")
            if spec_aug:
                kk.call("kk_specaug", dmem, B, T, H, self.rng, 20, hp.spec_augment_time_mask_max, hp.spec_augment_freq_mask_max,
                        hp.spec_augment_num_time_masks, hp.spec_augment_num_freq_masks, 0)
            if self.embed_bwd_sorted and d.var_bins <= 1024:
                kk.call("kk_bucket_embed_add_bwd_sorted", dmem, be_order, be_items, G[f"{VA}.pitch_embedding.weight"],
                        G[f"{VA}.energy_embedding.weight"], Nd, H, d.var_bins)
            else:
                kk.call("kk_bucket_embed_add_bwd", dmem, pidx, eidx, fmask, G[f"{VA}.pitch_embedding.weight"],
                        G[f"{VA}.energy_embedding.weight"], B, T, H, d.var_bins)

        if tail_mode:
            # the memory tail needs the cross-attention dK / dV of all layers and nothing else of the decoder backward: it runs on the
            # third stream beside layer 0's self-attention backward (1) or beside the input projection's backward (2), or on the side
            # stream behind the encoder's backward (3)
            if tail_mode == 2:
                tail_fork = self._fork_point()
            input_projection_bwd()
            if tail_mode not in (3, 4):
                with self._on_stream(self._kv, "kv.", after=tail_fork):
                    memory_tail()
                self._join(self._kv)
        else:
            input_projection_bwd()
            memory_tail()
        self._mark("main: backward tail done")
        side_backward(fork)
        self._join(self._side)
        self._reduce_partials((B, T, Pn, Tp))
        self._comm_bucket("tail")                       # everything that was not a layer's weight matrix
        self._comm_join()
        self._mark("backward joined, partials reduced")
        if self._ss_step is not None:                   # the tile records of this micro-batch: what an optimizer boundary right behind it may use
            if self._ss_step["covered"]:
                self._ss_ready = (frozenset(self._ss_step["covered"]), int(self._ss_step["count"].value))
            self._ss_step = None
        if self._ow_seen is not None:                   # what this cycle's first micro-batch overwrote (see _zero_grad_step)
            seen = frozenset(self._ow_seen)
            if self._ow_expect is not None and seen != self._ow_expect:
                raise RuntimeError("the step zero-filled the gradient arena by a stale overwrite record: "
                                   f"{len(self._ow_expect - seen)} tensors were skipped but not overwritten")
            self._ow_sets[self._ow_key()] = seen
            self._ow_seen = None
        return out

    # ------------------------------------------------------------------ inference
    @torch.no_grad()
    def generate(self, ids: torch.Tensor, stress: Optional[torch.Tensor] = None, max_len: int = 4000,
                 stop_threshold: float = 0.5, min_len_ratio: float = 0.7, min_len_floor: int = 12,
                 max_len_ratio: float = 3.0, max_len_cap: int = 1600, post_expected_stop_threshold: float = 0.2,
                 check_every: int = 16, decode_graph: bool = True) -> torch.Tensor:
        """KokoroModel.forward_inference (model/model.py:676-790) + KokoroGenerator.generate (model/generator.py:24-127):
        encode, expand by the model's own durations, pick the pitch / energy embeddings from its own (clamped)
        predictions (variance_predictor.py:338-439 without targets), then decode one mel frame at a time against a KV
        cache until the stop head fires, the output goes quiet or the length bound is reached.  Returns [B, frames, mel]
        clamped to [-11.5, 2] like the reference.  Weights: the engine's current parameters, dropout off.

        Decoder step = the training kernels at Sq = 1: the self-attention cache holds the NORMALISED (and, for K,
        rotated by the absolute position) heads time-major as [t][B*H], so one attention launch with B*heads "heads"
        serves the whole batch; the cross-attention K|V of all layers come from one GEMM (_cross_kv_fwd_all).  Like
        the reference, the single query of a step is rotated by position 0 (RoPE offsets default to 0 in its
        incremental path, transformers.py:276).  The stop decision needs the host; the reference syncs every frame,
        here the frames of `check_every` steps are decoded before the host looks (frames past the stop are dropped, so
        the result is the same).  decode_graph: frames 1.. are replays of ONE hipGraph of the step, captured after frame 0 ran
        eagerly (the step's launches index everything by a device-side frame counter); False issues the same launches one by one."""
        d, P, H, M, h = self.dims, self.arena.P, self.dims.hidden, self.dims.mel, self.dims.heads
        VA = "duration_adaptor.variance_adaptor"
        ids = ids.to(self.device, torch.int64).contiguous()
        stress = stress.to(self.device, torch.int64).contiguous() if stress is not None else None
        B, Pn = ids.shape
        Ne = B * Pn
        edt, ddt = self.enc_dt, self.dec_dt
        pe = P["positional_encoding.pe"].view(d.max_len, H)
        saved_drop, self.train_dropout = self.train_dropout, False
        try:
            # ---- encode_text (model.py:375-388) ----
            text_mask = self._buf("gen.text_mask", B, Pn, dtype=torch.uint8)
            kk.call("kk_ids_eq_zero", ids, text_mask, Ne)
            x = self._buf("enc.x0", Ne, H)
            kk.call("kk_embed_fwd", ids, stress, P["text_embedding.weight"], P["stress_embedding.weight"] if stress is not None else None,
                    pe, x, B, Pn, H, float(H ** 0.5), self.rng, 1, 0.0)
            y1 = None
            for i in range(d.enc_layers):
                pf, key = f"transformer_encoder_layers.{i}", f"enc{i}"
                if y1 is None:
                    y1 = self._ln_fwd(key + ".ln1", x, pf + ".norm1", edt)
                xm = self._buf(key + ".xm", Ne, H)
                y2 = self._attn_fwd(key + ".sa", pf + ".self_attn", y1, None, B, Pn, Pn, True, False, text_mask, x, xm,
                                    next_ln=(key + ".ln2", pf + ".norm2", edt))
                xo = self._buf(key + ".xo", Ne, H)
                nxt = ((f"enc{i + 1}.ln1", f"transformer_encoder_layers.{i + 1}.norm1", edt) if i + 1 < d.enc_layers
                       else ("enc.norm", "encoder_norm", torch.float32))
                y1 = self._ffn_fwd(key + ".ff", pf + ".ff", y2, xm, xo, d.enc_ff, Pn, next_ln=nxt)
                x = xo
            enc = y1
            # ---- variance adaptor without targets ----
            log_dur = self._buf("out.log_dur", B, Pn)
            col_e = self._buf("vp.col_enc", Ne, 3 * H, dtype=edt)
            kk.call("kk_im2col3_fwd", enc, col_e, B, Pn, H, CHUNK, _b16(col_e))
            self._varpred_fwd("vp.dur", f"{VA}.duration_predictor", enc, col_e, B, Pn, text_mask, log_dur)
            dur = torch.clamp(torch.round(torch.expm1(log_dur)), min=0).to(torch.int64)
            expected = max(int(dur.sum(dim=1).max()), 3)           # (host sync: the expanded length sizes everything below)
            T = expected
            if T > d.max_len:
                raise ValueError(f"predicted length {T} exceeds the positional table ({d.max_len})")
            Nd = B * T
            idx, lens, tot = (self._buf("lr.idx", B, T, dtype=torch.int64), self._buf("lr.lens", B, dtype=torch.int64),
                              self._buf("lr.total", B, dtype=torch.int64))
            kk.call("kk_length_regulate_index", dur, idx, lens, tot, B, Pn, T)
            xf = self._buf("va.xf", Nd, H)
            kk.call("kk_length_regulate_gather", enc, idx, xf, B, Pn, T, H)
            fmask = (torch.arange(T, device=self.device)[None, :] >= lens[:, None]).to(torch.uint8).contiguous()
            col_f = self._buf("vp.col_frames", Nd, 3 * H, dtype=ddt)
            kk.call("kk_im2col3_fwd", xf, col_f, B, T, H, CHUNK, _b16(col_f))
            pitch, energy = self._buf("out.pitch", B, T), self._buf("out.energy", B, T)
            self._varpred_fwd("vp.pitch", f"{VA}.pitch_predictor", xf, col_f, B, T, fmask, pitch)
            self._varpred_fwd("vp.energy", f"{VA}.energy_predictor", xf, col_f, B, T, fmask, energy)
            memory, fm2 = self._buf("va.memory", Nd, H, dtype=ddt), self._buf("va.fmask", B, T, dtype=torch.uint8)
            pidx, eidx = self._buf("va.pidx", B, T, dtype=torch.int32), self._buf("va.eidx", B, T, dtype=torch.int32)
            kk.call("kk_bucket_embed_add_fwd", xf, pitch.clamp(0.0, 1.0), energy.clamp(0.0, 1.0), P[f"{VA}.pitch_bins"],
                    P[f"{VA}.energy_bins"], P[f"{VA}.pitch_embedding.weight"], P[f"{VA}.energy_embedding.weight"], lens, memory,
                    pidx, eidx, fm2, B, T, H, d.var_bins, _b16(memory))
            self._cross_kv_fwd_all(memory, Nd, T, ddt)             # cross-attention K|V of every layer, normalised
            # ---- generation bounds (model.py:741-750) ----
            min_expected = max(min_len_floor, int(expected * min_len_ratio))
            max_expected = min(max_len, max(expected + 80, int(expected * max_len_ratio)), max_len_cap)
            if max_expected <= min_expected:
                max_expected = min(max_len, min_expected + 1)
            if max_expected > d.max_len:
                raise ValueError(f"generation bound {max_expected} exceeds the positional table ({d.max_len})")
            cos, sin = self._rope_tables(d.max_len)                                # (the whole tables: the step indexes them by t)
            BH, L1 = B * H, max_expected + 1
            Kc = [self._buf(f"gen.dec{i}.kcache", max_expected, BH, dtype=ddt) for i in range(d.dec_layers)]
            Vc = [self._buf(f"gen.dec{i}.vcache", max_expected, BH, dtype=ddt) for i in range(d.dec_layers)]
            for c_ in Kc + Vc:
                c_.zero_()                                                         # (rows >= t are masked, but 0 * garbage must stay 0)
            mel_out = self._buf("gen.mel", B, L1, M)                               # row 0 = the all-zero first input
            mel_out.zero_()
            stop_logit = self._buf("gen.stop", max_expected, B)
            y = self._buf("gen.y", B, H)
            # The step's launches take the frame index from device memory (kk_decode_*): their arguments are the same for every
            # frame, so ONE captured hipGraph of the step serves the whole utterance (decode_graph; ~100 launches per frame are
            # host-bound when issued one by one).  The self-attention runs over the whole cache under a key mask that opens key t.
            t_dev = self._buf("gen.t", 1, dtype=torch.int32)
            t_dev.zero_()
            kmask = self._buf("gen.kmask", 1, max_expected, dtype=torch.uint8)
            kmask.fill_(1)
            frame_in, frame_out, stop_now = self._buf("gen.frame_in", B, M), self._buf("gen.frame_out", B, M), self._buf("gen.stop_now", B)
            pe_row, cos_row, sin_row = self._buf("gen.pe_row", 1, H), self._buf("gen.cos_row", 1, 64), self._buf("gen.sin_row", 1, 64)

            def decode_step():
                kk.call("kk_decode_prologue", mel_out, frame_in, pe, pe_row, cos, sin, cos_row, sin_row, kmask, t_dev, B, L1, M, H)
                # mel_projection_in + positional encoding at offset t (model.py:541-545)
                self._linear(frame_in, self._W("mel_projection_in.weight"), P["mel_projection_in.bias"], y, res=pe_row, res_mod=1)
                n1 = self._ln_fwd("gen.dec0.ln1", y, "decoder.layers.0.norm1", ddt)
                yl = y
                for i in range(d.dec_layers):
                    pf, key = f"decoder.layers.{i}", f"gen.dec{i}"
                    gq, gk, gv = P[pf + ".self_attn.q_norm.weight"], P[pf + ".self_attn.k_norm.weight"], P[pf + ".self_attn.v_norm.weight"]
                    raw, nrm = self._buf(key + ".qkv_raw", B, 3 * H, dtype=ddt), self._buf(key + ".qkv_n", B, 3 * H, dtype=ddt)
                    self._proj_headnorm(n1, self._Wf(pf + ".self_attn.w_q.weight", 3), raw, nrm, 1, (gq, gk, gv), 2, cos_row, sin_row)
                    qb = self._buf(key + ".q", 1, BH, dtype=ddt)
                    kk.call("kk_decode_cache_append", nrm, qb, Kc[i], Vc[i], t_dev, B, H, _b16(nrm))
                    ctx, lse = self._buf(key + ".ctx", 1, BH, dtype=ddt), self._buf(key + ".lse", 1, B * h, 1)
                    kk.call("kk_attn_fwd", qb, Kc[i], Vc[i], ctx, lse, 1, B * h, 1, max_expected, BH, BH, BH, BH, kmask, 0, 0.125, self.rng,
                            0, 0.0, self.math, _b16(qb))
                    proj = self._buf("tmp.gen_proj", B, H)
                    self._linear(ctx.view(B, H), self._W(pf + ".self_attn.w_o.weight"), P[pf + ".self_attn.w_o.bias"], proj)
                    ya = self._buf(key + ".xa", B, H)
                    n2 = self._sublayer_tail(proj, yl, ya, 1, 0, 0.0, 0.0, 0.0, None, None, (key + ".ln2", pf + ".norm2", ddt))
                    yc = self._buf(key + ".xc", B, H)
                    n3 = self._attn_fwd(key + ".ca", pf + ".cross_attn", n2, memory, B, 1, T, False, False, fm2, ya, yc,
                                        next_ln=(key + ".ln3", pf + ".norm3", ddt), layer=i)
                    yo = self._buf(key + ".xo", B, H)
                    nxt = ((f"gen.dec{i + 1}.ln1", f"decoder.layers.{i + 1}.norm1", ddt) if i + 1 < d.dec_layers
                           else ("gen.dec.norm", "decoder.norm", ddt))
                    n1 = self._ffn_fwd(key + ".ff", pf + ".ff", n3, yc, yo, d.dec_ff, 1, next_ln=nxt)
                    yl = yo
                dec_out = n1
                self._linear(dec_out, self._W("mel_projection_out.weight"), P["mel_projection_out.bias"], frame_out)
                kk.call("kk_rowdot_fwd", dec_out, P["stop_token_predictor.weight"], P["stop_token_predictor.bias"], None, stop_now,
                        B, H, 1, 0, _b16(dec_out))
                kk.call("kk_decode_epilogue", frame_out, stop_now, mel_out, stop_logit, t_dev, B, L1, M)

            frames = max_expected
            done = 0
            step_graph = None
            for t in range(max_expected):
                if step_graph is not None:
                    step_graph.replay()
                else:
                    decode_step()                          # frame 0 eagerly: it sizes the workspaces a capture may not allocate
                    if decode_graph and max_expected > 1:
                        with self.capture_lock:
                            torch.cuda.synchronize()
                            step_graph = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(step_graph, capture_error_mode="thread_local"):
                                decode_step()
                if (t + 1) % check_every == 0 or t + 1 == max_expected:
                    sp = torch.sigmoid(stop_logit[done:t + 1]).mean(dim=1).cpu().tolist()
                    mel_host = mel_out[:, 1:t + 2].float().cpu() if t + 1 >= 30 else None
                    stop_at = None
                    for tt in range(done, t + 1):
                        if tt < min_expected:
                            continue
                        thr = stop_threshold if tt < expected else min(stop_threshold, post_expected_stop_threshold)
                        if sp[tt - done] > thr:
                            stop_at = tt
                            break
                        if tt + 1 >= 30 and float(mel_host[:, tt - 29:tt + 1].mean()) < -9.5:
                            stop_at = tt
                            break
                    done = t + 1
                    if stop_at is not None:
                        frames = stop_at + 1
                        break
            return mel_out[:, 1:frames + 1].clamp(min=-11.5, max=2.0).clone()
        finally:
            self.train_dropout = saved_drop

    # ------------------------------------------------------------------ optimizer boundary
    def _opt_cfg(self, mel_length: int) -> kk.KkOptCfg:
        hp = self.hp
        c = spec.lr_schedule_consts(hp, self.total_steps)
        return kk.KkOptCfg(c["learning_rate"], c["max_lr"], c["warmup_start_lr"], c["warmup_target_lr"], c["pct_start"],
                           c["div_factor"], c["final_div_factor"], c["warmup_steps"], c["onecycle_steps"], c["use_warmup"],
                           hp.adam_betas[0], hp.adam_betas[1], hp.adam_eps, hp.max_grad_norm, mel_length,
                           hp.grad_explosion_ema_alpha, hp.grad_explosion_abs_floor, hp.grad_explosion_multiplier,
                           hp.grad_explosion_warmup_floor, hp.grad_explosion_warmup_steps, hp.grad_explosion_min_ema_steps,
                           hp.ema_decay, hp.dec_ffn_max_weight_norm, max(1, int(hp.ema_update_every)),
                           0 if hp.use_onecycle_lr else 1,
                           0.0 if hp.use_onecycle_lr else spec.cosine_restart_factor(self.lr_epoch, hp.lr_T_0, hp.lr_T_mult),
                           hp.lr_eta_min)

    def zero_grad(self) -> None:
        self.arena.g.zero_()

    def _ow_key(self) -> Tuple:
        return (self.math, self.enc_dt, self.dec_dt, self.use_shadow, self.group_wgrads)

    def _zero_grad_step(self) -> None:
        """The zero-fill inside a step (forward_backward(zero_grads=True)): everything, or — once a step of this precision mode
        has been seen — everything but the tensors its grouped weight-gradient launches overwrite, as ONE kk_zero_many launch."""
        a = self.arena
        rec = self._ow_sets.get(self._ow_key()) if self.zero_skip_overwritten else None
        if not rec:
            self._ow_expect = None
            self.zero_grad()
            return
        self._ow_expect = rec

        def build():
            base, spans, pos = a.g.data_ptr(), [], 0
            for ptr, n in sorted(rec):
                b = (ptr - base) // 4
                if b < pos or b % 4 or n % 4:
                    raise RuntimeError("overlapping or misaligned weight-gradient tensors in the overwrite record")
                if b > pos:
                    spans.append(a.g[pos:b])
                pos = b + n
            if pos < a.total:
                spans.append(a.g[pos:a.total])
            return [kk.zero_table(spans[i:i + 160]) for i in range(0, len(spans), 160)]
        for dst, nbytes, n in self._table(("zero", a.g.data_ptr(), rec), build):
            kk.call("kk_zero_many", dst, nbytes, n)

    def optimizer_step(self, mel_length: int, tile_norms=None) -> None:
        """Pre-clip → total norm → non-finite skip / explosion tracker / adaptive + global clip → fused AdamW+EMA →
        FFN weight-norm projection.  All decisions on the device (see csrc/kk_optim.hip).
        tile_norms = (segments, records) from the micro-batch right in front (train_step passes _ss_ready; see grad_norm_from_wgrads):
        the norm pass skips those segments — only for callers that have not touched the gradient arena in between."""
        a, hp = self.arena, self.hp
        cfg = self._opt_cfg(mel_length)
        self._mark("optimizer start")
        skip, extra = None, 0
        if tile_norms is not None and tile_norms[1] > 0:
            skip = self._ss_masks.get(tile_norms[0])
            if skip is None and not torch.cuda.is_current_stream_capturing():
                m = torch.zeros(a.nseg, dtype=torch.int32)
                m[sorted(tile_norms[0])] = 1
                skip = self._ss_masks[tile_norms[0]] = m.to(self.device)
            extra = tile_norms[1] if skip is not None else 0
        # grad_sumsq: every segment stored once per call by kk_seg_sumsq (records merged in a fixed order: replicas agree bit for bit).
        # p_sumsq (fixed-point atomics) is private to this sequence: kk_opt_prepare, the one-workgroup launch in front of its writer,
        # leaves it zero — no zero-fill launch on the optimizer's chain.
        with self._acc_guard():
            sc = 1 if self.self_cleaning_acc else 0
            kk.call("kk_seg_sumsq", a.g, a.block_seg, a.nblocks, self.grad_sumsq, a.nseg, self.sumsq_ws, skip, extra)
            kk.call("kk_opt_prepare", self.grad_sumsq, a.seg_preclip, a.seg_lr_mult, a.seg_wd, a.nseg, self.max_dur, cfg,
                    self.opt_state, self.seg_gscale, self.seg_decay, self.seg_stepsize, self.step_consts,
                    None, self.p_sumsq if sc else None)
            kk.call("kk_adamw_ema", a.p, a.g, a.m, a.v, a.ema, a.block_seg, a.nblocks, self.seg_gscale, self.seg_decay,
                    self.seg_stepsize, a.seg_flags, self.step_consts, hp.adam_betas[0], hp.adam_betas[1], hp.ema_decay,
                    self.p_sumsq, a.nseg, a.p16, sc)
            kk.call("kk_weight_norm_project", a.p, a.block_seg, a.nblocks, self.p_sumsq, a.seg_flags, self.step_consts,
                    float(hp.dec_ffn_max_weight_norm), a.p16)
        self._mark("optimizer done")

    def train_step(self, batch: Dict[str, torch.Tensor], accumulation_divisor: Optional[int] = None,
                   boundary: Optional[bool] = None, grad_sync=None, expanded_len: Optional[int] = None) -> torch.Tensor:
        """One micro-batch of training in the reference's order (trainer.py:2257-2477): zero grads at the start of an
        accumulation cycle, forward+backward with loss_scale = adaptive/divisor, optimizer boundary when the cycle
        completes.  Returns the device tensor of 6 losses (no host sync)."""
        G = max(1, int(self.hp.gradient_accumulation_steps))
        div = accumulation_divisor if accumulation_divisor is not None else G
        if self.micro_in_cycle == 0:
            self.zero_grad()
        is_boundary = bool(boundary) if boundary is not None else self.micro_in_cycle + 1 >= G
        self._exchange_now = is_boundary                 # (in-step bucket exchange: only the boundary micro-batch communicates)
        self._first_micro = self.micro_in_cycle == 0          # (zeroed just above)
        self._external_sync = grad_sync is not None
        try:
            out = self.forward_backward(batch, loss_scale=self.dp_loss_scale / div, adaptive=True, expanded_len=expanded_len)
        finally:
            self._first_micro = False
            self._external_sync = False
        self._exchange_now = True
        self.micro_in_cycle += 1
        if is_boundary:
            if grad_sync is not None:
                grad_sync(self.arena.g)              # data parallel: SUM over ranks (dp.GradSync)
            self.optimizer_step(canonical_mel_length(self.global_mel_length, batch["mel_specs"].shape[1]),
                                self._ss_ready if grad_sync is None else None)
            self._ss_ready = None
            self.micro_in_cycle = 0
        return out["losses"]

    def train_step_auto(self, batch: Dict[str, torch.Tensor], accumulation_divisor: Optional[int] = None,
                        boundary: Optional[bool] = None, grad_sync=None, expanded_len: Optional[int] = None) -> torch.Tensor:
        """train_step for a stream of batches whose shapes may or may not repeat (the trainer's entry point): a shape that
        comes back is replayed from hipGraphs (train_step_graphed: ~0.2 ms of host time instead of ~2.5 ms for ~700
        launches), a shape seen for the first time runs eagerly.  Large dynamic batches (max_frames 16384: 15-20 ms of
        GPU time per step) hide the eager launches anyway; small fixed shapes are where the graphs pay.  At most
        max_graphs shapes are kept (least recently used first out)."""
        B, T = batch["mel_specs"].shape[:2]
        key = (B, T, batch["phoneme_indices"].shape[1], T if expanded_len is None else int(expanded_len))
        if self.loss_sync is None or getattr(self.loss_sync, "capturable", False):     # (a torch.distributed loss exchange is eager)
            if key in self._graphs:
                return self.train_step_graphed(batch, grad_sync, accumulation_divisor, boundary, expanded_len)
            n = self._shape_seen.get(key, 0) + 1
            self._shape_seen[key] = n
            self._shape_seen.move_to_end(key)
            if len(self._shape_seen) > 4096:
                self._shape_seen.popitem(last=False)
            if n >= 2:
                return self.train_step_graphed(batch, grad_sync, accumulation_divisor, boundary, expanded_len)
        return self.train_step(batch, accumulation_divisor, boundary, grad_sync, expanded_len)

    # ------------------------------------------------------------------ hipGraph replay of a whole step
    def _capture_fb(self, static, scale, first, expanded_len) -> "torch.cuda.CUDAGraph":
        """Capture one micro-batch variant (forward + losses + backward) of a batch shape; called under capture_lock."""
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: with an RCCL process group alive, its watchdog thread polls events while we capture; only
        # this thread's calls must be capture-safe
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self.forward_backward(static, loss_scale=scale, adaptive=True, zero_grads=first, expanded_len=expanded_len)
        return g

    def train_step_graphed(self, batch: Dict[str, torch.Tensor], grad_sync=None, accumulation_divisor: Optional[int] = None,
                           boundary: Optional[bool] = None, expanded_len: Optional[int] = None) -> torch.Tensor:
        """Same semantics as train_step (gradient accumulation included), but the kernel sequence of a micro-batch is
        captured once per (batch shape, accumulation divisor, first-of-cycle) into hipGraphs and replayed (a step is ~700
        launches; eager launch overhead would dominate), and the optimizer boundary once per (shape, mel length).
        `grad_sync` (data parallel) runs between the backward graph and the optimizer graph.  The first call with a new
        shape runs eagerly: it sizes the workspace and builds the descriptor tables, neither of which may happen inside
        a capture."""
        B, T = batch["mel_specs"].shape[:2]
        Tp = T if expanded_len is None else int(expanded_len)
        key = (B, T, batch["phoneme_indices"].shape[1], Tp)
        if self.loss_sync is not None and not getattr(self.loss_sync, "capturable", False):
            raise RuntimeError("train_step_graphed: this loss_sync exchanges the loss counts through torch.distributed between "
                               "kernels of the step, which a hipGraph cannot hold; use train_step, or the RCCL exchange "
                               "(dp.BucketedExchange, backend 'rccl'), whose collectives are captured with the step")
        G = max(1, int(self.hp.gradient_accumulation_steps))
        div = int(accumulation_divisor) if accumulation_divisor is not None else G
        first = self.micro_in_cycle == 0
        is_boundary = bool(boundary) if boundary is not None else self.micro_in_cycle + 1 >= G
        scale = self.dp_loss_scale / div
        mel_length = canonical_mel_length(self.global_mel_length, T)
        ent = self._graphs.get(key)
        self._exchange_now = is_boundary
        if ent is None:                               # first sight of a shape: eager (allocates the workspaces and the
            static = {k: v.clone() for k, v in batch.items()}      # reduction tables — host-to-device copies, illegal in a capture)
            self._external_sync = grad_sync is not None
            try:
                self._fb(static, scale, True, True, first, expanded_len)
            finally:
                self._external_sync = False
            self._exchange_now = True
            # (registered only now: growing a buffer during the eager pass drops every graph entry)
            if len(self._graphs) >= self.max_graphs:
                torch.cuda.synchronize(self.device)   # a graph about to be destroyed may still be running
                self._graphs.popitem(last=False)
            self._graphs[key] = {"static": static, "fb": {}, "opt": {}}
            self.micro_in_cycle += 1
            if is_boundary:
                if grad_sync is not None:
                    grad_sync(self.arena.g)
                self.optimizer_step(mel_length, self._ss_ready if grad_sync is None else None)
                self._ss_ready = None
                self.micro_in_cycle = 0
            return self.losses
        self._graphs.move_to_end(key)
        static = ent["static"]
        moved = []
        for k, v in batch.items():
            d = static[k]
            if v.data_ptr() == d.data_ptr():
                continue
            if v.shape == d.shape and v.dtype == d.dtype and v.device == d.device and v.is_contiguous() and d.is_contiguous():
                moved.append((d, v))
            else:
                d.copy_(v, non_blocking=True)
        if moved:
            kk.copy_many(moved)                        # one launch for the whole batch
        # (kk_losses_finalize takes the GLOBAL mel length by value — the adaptive loss scale depends on it above 1400 frames — so a
        # graph captured under one value must not be replayed under another: ADVICE r3; the CANONICAL value, so that a ragged
        # data-parallel run replays one capture per local shape for every global length up to 1400: canonical_mel_length)
        fkey = (div, first, self.train_dropout, self.spec_augment_active, self.math, self.dp_comm is not None and is_boundary,
                mel_length if self.loss_sync is not None else 0, self.loss_sync is not None, self.self_cleaning_acc, (self.attn_keep_gen, self.attn_keep_gen_rate),
                self.grad_norm_from_wgrads and grad_sync is None)
        with self._acc_guard():                        # (a replayed graph assumes the accumulators' zero state like an eager step)
            fbe = ent["fb"].get(fkey)
            if fbe is None:
                with self.capture_lock:
                    self._external_sync = grad_sync is not None
                    try:
                        g_ = self._capture_fb(static, scale, first, expanded_len)
                    finally:
                        self._external_sync = False
                    fbe = ent["fb"][fkey] = (g_, self._ss_ready)      # (the tile-norm records this launch sequence leaves: the same at every replay)
            self._exchange_now = True
            fbe[0].replay()
            self._ss_ready = fbe[1]
        self.micro_in_cycle += 1
        if grad_sync is not None and is_boundary:
            grad_sync(self.arena.g)
        if is_boundary:
            # (KkOptCfg travels by value: the legacy schedule's factor changes once per epoch -> one re-capture per epoch; the graphs of
            #  earlier epochs are dropped then — they can never be replayed again and would otherwise pile up, epochs x shapes: ADVICE r5)
            epoch = 0 if self.hp.use_onecycle_lr else self.lr_epoch
            tn = self._ss_ready if grad_sync is None else None
            self._ss_ready = None
            if tn is not None and tn[0] not in self._ss_masks:       # (its skip table is built by an eager boundary only: no copies under capture)
                tn = None
            okey = (mel_length, epoch, self.self_cleaning_acc, tn)
            for k in [k for k in ent["opt"] if k[1] != epoch]:
                torch.cuda.synchronize(self.device)   # (a graph about to be destroyed may still be running)
                del ent["opt"][k]
            with self._acc_guard():
                opt = ent["opt"].get(okey)
                if opt is None:
                    with self.capture_lock:
                        torch.cuda.synchronize()
                        opt = ent["opt"][okey] = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(opt, capture_error_mode="thread_local"):
                            self.optimizer_step(mel_length, tn)
                opt.replay()
            self.micro_in_cycle = 0
        return self.losses

    def encoder_stack_error(self) -> int:
        """Non-zero when a group barrier of kk_encoder_stack_fwd has timed out since the last reset (host read: synchronises)."""
        return int(self._enc_sync[0].item())

    def check_encoder_stack(self) -> None:
        """Raise if a group barrier of the persistent encoder launch has timed out.  kk_encoder_stack_fwd is a plain launch whose
        256 workgroups must be co-resident; every spin is bounded, and a workgroup that gives up sets word 0 of the sync buffer
        and carries on with stale activations — the step (and every later one: the word makes later barriers fall through) is
        then garbage that no finite-value guard sees.  The trainer calls this wherever it synchronises anyway (end of an epoch,
        before validation, before a checkpoint is written).  Handling it = clear the word (so later launches wait again), fall
        back to the per-kernel encoder for the rest of the run, drop the graphs that hold the launch, and fail the caller: the
        weights since the last check cannot be trusted."""
        # data parallel: the ranks decide TOGETHER — a rank that raised alone would leave the others waiting in the next
        # collective.  The check sits at the trainer's sync points (epoch end, validation, checkpoint), which every rank reaches, and
        # EVERY rank enters the all-reduce — also one whose own fused launch is already off (it contributes 0; ADVICE r4: returning
        # early there left the other ranks alone in the collective).
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not self.enc_fused and not multi:
            return
        code = self.encoder_stack_error() if self.enc_fused else 0
        if multi:
            word = torch.tensor([code], dtype=torch.int64, device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(word, op=dist.ReduceOp.MAX)
            code = int(word.item())
        if code:
            self._enc_sync.zero_()
            if self.enc_fused:
                self.enc_fused = False
                self._graphs.clear()
            raise RuntimeError(f"kk_encoder_stack_fwd: a group barrier timed out (code {code}); the activations of at least one "
                               "step were stale.  The fused encoder launch is now off for this engine (per-kernel sequence); "
                               "reload the last checkpoint before continuing.")

    def opt_stats(self) -> Dict[str, float]:
        """Host read-back of the device optimizer state (synchronises; for logging/tests only)."""
        s = self.opt_state.cpu().tolist()
        return {k.lower(): s[v] for k, v in kk.OS.items() if k != "SIZE"}
