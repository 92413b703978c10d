"""Cached-feature reader, collate and batch samplers.

Reads the per-utterance `.pt` files the reference's `kokoro-precompute` writes under `{corpus}/.feature_cache/`
(schema v7, reference data/dataset.py:849-862: mel_spec [80,T] f32, phoneme_indices/stress_indices/phoneme_durations
[P] i64, stop_token_targets/pitch/energy [T] f32, mel_length, phoneme_length, text, audio_file, _cache_version).
The audio front-end, MFA alignment and the phonemizer that PRODUCE these files stay on the reference (out of scope).
`collate_fn` reproduces the reference's zero-padded batch dict (data/dataset.py:871-921).  The frame-budget sampler
restates the reference's DynamicFrameBatchSampler (dataset.py:924-1147: quantile buckets, greedy packing under
batch_size * longest <= max_frames, heavy-batch spreading) and is pinned against it by tests/golden/sampler.json.
"""
from __future__ import annotations

import json
import os
import random
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Sequence

import torch

FEATURE_CACHE_VERSION = 7
_TENSOR_KEYS = ("mel_spec", "phoneme_indices", "stress_indices", "phoneme_durations", "stop_token_targets", "pitch", "energy")


def clip_durations(dur: torch.Tensor, T: int) -> torch.Tensor:
    """Durations of an utterance whose mel THIS READER cut to T frames (a cache built with a larger max_seq_length than the
    run's): every phoneme keeps the frames that fall before the cut, so sum(dur) == min(sum(dur), T) and the expansion is a
    prefix of the original one; phonemes past the cut get duration 0 (the duration loss masks them: losses.py d > 0).

    DELIBERATE DIVERGENCE, not the reference's rule.  The reference clips only when it BUILDS a sample (kokoro-precompute,
    data/dataset.py:704-707) and reconciles there by `dur[-1] = max(1, dur[-1] + (T - sum))` followed by `clamp(min=1)`
    (:769-776), which keeps every duration >= 1 and — when the last phoneme is shorter than the cut — leaves sum(dur) > T
    (the model then expands to T' > T frames, model.py:607-628).  It never re-clips a cached file, so a cache whose files are
    longer than the run's max_seq_length has no reference behaviour to copy; here the frames kept are exactly the frames whose
    mel survives.  `reference_reconcile` below restates the reference's rule for the tests that pin the difference."""
    cum = torch.cumsum(dur.clamp(min=0), 0).clamp(max=T)
    return torch.diff(cum, prepend=cum.new_zeros(1))


def reference_reconcile(dur: torch.Tensor, T: int) -> torch.Tensor:
    """data/dataset.py:769-776 as written (used at precompute time by the reference; test infrastructure here)."""
    d = dur.clone()
    diff = int(T) - int(d.sum())
    if diff != 0 and len(d) > 0:
        d[-1] = max(1, int(d[-1]) + diff)
    return d.clamp(min=1)


def scan_cache(cache_dir: str) -> List[Dict]:
    """[{file, audio_length, phoneme_length}] for every cached utterance, sorted by length (dataset.py:398).  Reading the
    lengths means un-pickling every file once; the result is kept in a sidecar index next to the cache (file name, size,
    mtime, lengths) so that later runs — and the train / validation views of one run — do not repeat that."""
    d = Path(cache_dir)
    files = sorted(d.glob("*.pt"))
    if not files:
        raise FileNotFoundError(f"no cached features (*.pt) under {d}; run the reference's kokoro-precompute")
    index_path = d / ".kk_index.json"
    known = {}
    try:
        idx = json.loads(index_path.read_text())
        if idx.get("version") == FEATURE_CACHE_VERSION and idx.get("index_format") == 2:     # an index of another cache schema (or of
            known = {e["name"]: e for e in idx["entries"]}                                  # second-resolution mtimes) is discarded
    except Exception:
        known = {}
    metas, entries, dirty = [], [], False
    for f in files:
        st = f.stat()
        e = known.get(f.name)
        if e is None or e.get("size") != st.st_size or e.get("mtime_ns") != st.st_mtime_ns or e.get("cache_version") != FEATURE_CACHE_VERSION:
            it = torch.load(f, map_location="cpu", weights_only=False)
            ver = it.get("_cache_version")
            if ver != FEATURE_CACHE_VERSION:
                raise RuntimeError(f"{f.name}: feature cache version {ver}, expected {FEATURE_CACHE_VERSION}")
            e = {"name": f.name, "size": st.st_size, "mtime_ns": st.st_mtime_ns, "cache_version": int(ver),
                 "mel_length": int(it["mel_length"]), "phoneme_length": int(it["phoneme_length"])}
            dirty = True
        entries.append(e)
        metas.append({"file": f, "audio_length": e["mel_length"], "phoneme_length": e["phoneme_length"]})
    if dirty:
        try:
            tmp = index_path.with_suffix(".tmp%d" % os.getpid())
            tmp.write_text(json.dumps({"version": FEATURE_CACHE_VERSION, "index_format": 2, "entries": entries}))
            os.replace(tmp, index_path)
        except OSError:
            pass                                                   # read-only cache directory: scan again next time
    metas.sort(key=lambda m: m["audio_length"])                    # dataset.py:398 (sorted by length; stable)
    return metas


class CachedFeatureDataset:
    def __init__(self, cache_dir: str, indices: Optional[Sequence[int]] = None, max_seq_length: int = 1800,
                 memory_cache: bool = True, metas: Optional[List[Dict]] = None):
        self.dir = Path(cache_dir)
        metas = scan_cache(cache_dir) if metas is None else list(metas)
        # the reference's metadata holds the CLIPPED length (dataset.py:325-327) and sorts by it
        metas = sorted((dict(m, audio_length=min(int(m["audio_length"]), int(max_seq_length))) for m in metas),
                       key=lambda m: m["audio_length"])
        if indices is not None:
            metas = [metas[i] for i in indices if i < len(metas)]  # dataset.py:403-405
        self.samples = metas
        self.max_seq_length = max_seq_length
        self._mem: Optional[Dict[int, Dict]] = {} if memory_cache else None

    def __len__(self) -> int:
        return len(self.samples)

    def __getitem__(self, i: int) -> Dict:
        if self._mem is not None and i in self._mem:
            return self._mem[i]
        it = torch.load(self.samples[i]["file"], map_location="cpu", weights_only=False)
        for k in _TENSOR_KEYS:
            if k not in it:
                raise KeyError(f"{self.samples[i]['file'].name}: missing field {k}")
        T = min(int(it["mel_length"]), self.max_seq_length)
        if it["mel_spec"].shape[1] != T:                           # clip over-long utterances (dataset.py:704-707)
            it = dict(it)
            it["mel_spec"] = it["mel_spec"][:, :T]
            for k in ("stop_token_targets", "pitch", "energy"):
                it[k] = it[k][:T]
            it["phoneme_durations"] = clip_durations(it["phoneme_durations"], T)
            it["mel_length"] = T
        if self._mem is not None:
            self._mem[i] = it
        return it


def collate_fn(batch: List[Dict]) -> Dict[str, torch.Tensor]:
    B = len(batch)
    mel_len = [int(it["mel_length"]) for it in batch]
    ph_len = [int(it["phoneme_length"]) for it in batch]
    T, P, M = max(mel_len), max(ph_len), batch[0]["mel_spec"].shape[0]
    out = {"mel_specs": torch.zeros(B, T, M), "pitches": torch.zeros(B, T), "energies": torch.zeros(B, T),
           "stop_token_targets": torch.zeros(B, T), "phoneme_indices": torch.zeros(B, P, dtype=torch.long),
           "phoneme_durations": torch.zeros(B, P, dtype=torch.long), "stress_indices": torch.zeros(B, P, dtype=torch.long)}
    for i, it in enumerate(batch):
        t, p = mel_len[i], ph_len[i]
        out["mel_specs"][i, :t] = it["mel_spec"].T[:t]
        out["pitches"][i, :t] = it["pitch"][:t]
        out["energies"][i, :t] = it["energy"][:t]
        out["stop_token_targets"][i, :t] = it["stop_token_targets"][:t]
        out["phoneme_indices"][i, :p] = it["phoneme_indices"][:p]
        out["phoneme_durations"][i, :p] = it["phoneme_durations"][:p]
        out["stress_indices"][i, :p] = it["stress_indices"][:p]
    out["mel_lengths"] = torch.tensor(mel_len, dtype=torch.long)
    out["phoneme_lengths"] = torch.tensor(ph_len, dtype=torch.long)
    return out


# ---- the same batch, written straight into caller-provided (pinned) memory ------------------------------------------
# collate_fn above is the reference's contract and what validation / tests use.  The train loop's loader thread needs the
# batch in ONE pinned staging buffer and has a few milliseconds per batch, so it lays the nine tensors out in a flat byte
# range (batch_layout) and fills them with plain numpy copies (collate_into): no intermediate tensors, no torch dispatch
# (a torch op on a 128-core host wakes an OpenMP team for a 160 KB copy), per-utterance arrays prepared once.
BATCH_KEYS = ("mel_specs", "pitches", "energies", "stop_token_targets", "phoneme_indices", "phoneme_durations",
              "stress_indices", "mel_lengths", "phoneme_lengths")


def sample_arrays(it: Dict) -> Dict:
    """numpy views of one cached utterance, the mel already frame-major [T, M] (kept inside the item: built once)."""
    a = it.get("_np")
    if a is None:
        import numpy as np
        mel_T = np.ascontiguousarray(it["mel_spec"].numpy().T)
        it["mel_spec"] = torch.from_numpy(mel_T).T                 # same memory: the [M, T] tensor is now a view of it
        a = it["_np"] = {"mel": mel_T, "pitch": it["pitch"].numpy(), "energy": it["energy"].numpy(),
                         "stop": it["stop_token_targets"].numpy(), "ids": it["phoneme_indices"].numpy(),
                         "dur": it["phoneme_durations"].numpy(), "stress": it["stress_indices"].numpy()}
    return a


def batch_layout(items: List[Dict], max_mel: int = 1 << 30, max_ph: int = 1 << 30, align: int = 256):
    """(B, T, P, M, mel_len, ph_len, plan, total_bytes) of the padded batch of `items`, sequence dimensions capped like
    _cap_batch_sequence_dimensions (reference trainer.py:3364-3411); plan = [(key, byte offset, bytes, shape, dtype name)]."""
    B = len(items)
    mel_len = [min(int(it["mel_length"]), max_mel) for it in items]
    ph_len = [min(int(it["phoneme_length"]), max_ph) for it in items]
    T, P, M = max(mel_len), max(ph_len), int(items[0]["mel_spec"].shape[0])
    shapes = {"mel_specs": ((B, T, M), "float32"), "pitches": ((B, T), "float32"), "energies": ((B, T), "float32"),
              "stop_token_targets": ((B, T), "float32"), "phoneme_indices": ((B, P), "int64"),
              "phoneme_durations": ((B, P), "int64"), "stress_indices": ((B, P), "int64"), "mel_lengths": ((B,), "int64"),
              "phoneme_lengths": ((B,), "int64")}
    off, plan = 0, []
    for k in BATCH_KEYS:
        shape, dt = shapes[k]
        n = (4 if dt == "float32" else 8)
        for v in shape:
            n *= v
        plan.append((k, off, n, shape, dt))
        off += (n + align - 1) // align * align
    return B, T, P, M, mel_len, ph_len, plan, off


def collate_into(items: List[Dict], out: Dict, mel_len: List[int], ph_len: List[int]) -> None:
    """Fill the numpy arrays out[key] (shapes of batch_layout) with the zero-padded batch: equals
    cap_batch(collate_fn(items)) element for element."""
    for i, it in enumerate(items):
        a = sample_arrays(it)
        t, p = mel_len[i], ph_len[i]
        out["mel_specs"][i, :t] = a["mel"][:t]
        out["mel_specs"][i, t:] = 0.0
        for key, src in (("pitches", "pitch"), ("energies", "energy"), ("stop_token_targets", "stop")):
            out[key][i, :t] = a[src][:t]
            out[key][i, t:] = 0.0
        for key, src in (("phoneme_indices", "ids"), ("phoneme_durations", "dur"), ("stress_indices", "stress")):
            out[key][i, :p] = a[src][:p]
            out[key][i, p:] = 0
    out["mel_lengths"][:] = mel_len
    out["phoneme_lengths"][:] = ph_len


def split_indices(n: int, val_split: float):
    """90/10 split exactly as the reference draws it (training/trainer.py:284-296)."""
    idx = list(range(n))
    if val_split <= 0:
        return idx, []
    random.seed(42)
    random.shuffle(idx)
    k = int(n * (1 - val_split))
    return idx[:k], idx[k:]


def shard_steps(global_batches: List[List[int]], rank: int, world: int) -> List[List[int]]:
    """Step s of a data-parallel run = batches [s*world, (s+1)*world) of the deterministic global list, rank r takes the
    r-th.  The ragged tail (< world batches) is dropped so every rank runs the same number of steps —
    a rank with one step fewer would leave the others waiting in the gradient all-reduce."""
    if world <= 1:
        return global_batches
    n = len(global_batches) // world * world
    return global_batches[:n][rank::world]


def step_groups(global_batches: List[List[int]], world: int) -> List[List[List[int]]]:
    """The `world` batches that make up each data-parallel step (every rank can enumerate them: no communication)."""
    world = max(1, world)
    n = len(global_batches) // world * world
    return [global_batches[i:i + world] for i in range(0, n, world)]


class FixedBatchSampler:
    def __init__(self, n: int, batch_size: int, shuffle: bool = True, rank: int = 0, world: int = 1, seed: int = 0):
        self.n, self.bs, self.shuffle, self.rank, self.world, self.seed, self.epoch = n, batch_size, shuffle, rank, world, seed, 0

    def global_batches(self) -> List[List[int]]:
        order = list(range(self.n))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(order)
        return [order[i:i + self.bs] for i in range(0, self.n, self.bs)]

    def batches(self) -> List[List[int]]:
        return shard_steps(self.global_batches(), self.rank, self.world)

    def __len__(self) -> int:
        return len(self.batches())

    def __iter__(self) -> Iterator[List[int]]:
        yield from self.batches()


def length_based_batch_sampler(dataset, batch_size: int, shuffle: bool = True, rank: int = 0, world: int = 1, seed: int = 0,
                               drop_last: bool = False) -> "FrameBudgetBatchSampler":
    """The reference's LengthBasedBatchSampler (data/dataset.py:1150-1180): fixed batch size, samples grouped by length —
    the dynamic sampler with a frame budget that never binds and min_batch_size 1."""
    longest = max((int(m["audio_length"]) for m in dataset.samples), default=10000)
    return FrameBudgetBatchSampler(dataset, longest * batch_size, 1, batch_size, shuffle, rank, world, seed, drop_last)


class FrameBudgetBatchSampler:
    """The reference's DynamicFrameBatchSampler (data/dataset.py:924-1147), restated, plus rank sharding.

    Batches are lists of dataset indices whose cost `len(batch) * longest_sample_frames` stays within `max_frames`
    (and whose size stays within `max_batch_size`):
      1. quantile buckets — `min(16, max(1, int(sqrt(N))))` buckets cut at the percentiles of the sample lengths, a
         sample goes to the last bucket whose lower cut is <= its length (dataset.py:1027-1045);
      2. inside each bucket: shuffle, then greedy packing in that order; a batch below `min_batch_size` is kept
         unless `drop_last` (dataset.py:1049-1075);
      3. heavy-batch spreading — the `max(2, int(sqrt(n)))` costliest batches are anchors placed at even distances
         (heaviest first), the shuffled remaining batches fill the gaps (dataset.py:1089-1127).
    Randomness: the reference draws from the process-global `random` module; here every epoch draws from
    `random.Random(seed + epoch)` in the same order (one shuffle per non-empty bucket, then one for the light
    batches), so all ranks build the identical list without communication and a globally seeded reference run
    reproduces it (tests/golden/sampler.json)."""

    def __init__(self, dataset, max_frames: int = 20000, min_batch_size: int = 4, max_batch_size: int = 32,
                 shuffle: bool = True, rank: int = 0, world: int = 1, seed: int = 0, drop_last: bool = False):
        self.ds, self.max_frames, self.min_bs, self.max_bs = dataset, max_frames, min_batch_size, max_batch_size
        self.shuffle, self.rank, self.world, self.seed, self.epoch, self.drop_last = shuffle, rank, world, seed, 0, drop_last
        self._cached = (None, None)                                # (epoch, batch list): len() and batches() share one build

    def _frames(self, i: int) -> int:
        return int(self.ds.samples[i]["audio_length"])

    def global_batches(self) -> List[List[int]]:
        if self._cached[0] != self.epoch:
            self._cached = (self.epoch, self._build())
        return self._cached[1]

    def _build(self) -> List[List[int]]:
        import numpy as np
        n_samples = len(self.ds.samples)
        if n_samples == 0:
            return []
        rng = random.Random(self.seed + self.epoch)
        lengths = np.array([self._frames(i) for i in range(n_samples)], dtype=np.int64)
        n_buckets = min(16, max(1, int(np.sqrt(n_samples))))
        cuts = np.percentile(lengths, np.linspace(0, 100, n_buckets + 1))
        buckets: List[List[int]] = [[] for _ in range(n_buckets)]
        for i, ln in enumerate(lengths.tolist()):
            b = int(np.searchsorted(cuts, ln, side="right") - 1)
            buckets[max(0, min(n_buckets - 1, b))].append(i)
        out: List[List[int]] = []
        keep = lambda batch: bool(batch) and (len(batch) >= self.min_bs or not self.drop_last)
        for bucket in buckets:
            if not bucket:
                continue
            if self.shuffle:
                rng.shuffle(bucket)
            cur: List[int] = []
            longest = 0
            for i in bucket:
                fr = self._frames(i)
                if cur and ((len(cur) + 1) * max(longest, fr) > self.max_frames or len(cur) >= self.max_bs):
                    if keep(cur):
                        out.append(cur)
                    cur, longest = [], 0
                cur.append(i)
                longest = max(longest, fr)
            if keep(cur):
                out.append(cur)
        if self.shuffle and len(out) > 1:
            n = len(out)
            n_heavy = max(2, int(n ** 0.5))
            cost = [max((self._frames(i) for i in b), default=0) * len(b) for b in out]
            ranked = [out[i] for i in sorted(range(n), key=lambda i: cost[i], reverse=True)]     # stable, like the reference
            heavy, light = ranked[:n_heavy], ranked[n_heavy:]
            rng.shuffle(light)
            gap, rem = divmod(len(light), n_heavy)
            res: List[List[int]] = []
            start = 0
            for k, anchor in enumerate(heavy):
                end = start + gap + (1 if k < rem else 0)
                res.append(anchor)
                res.extend(light[start:end])
                start = end
            out = res
        elif self.shuffle:
            rng.shuffle(out)
        return out

    def batches(self) -> List[List[int]]:
        return shard_steps(self.global_batches(), self.rank, self.world)

    def __len__(self) -> int:
        return len(self.batches())

    def __iter__(self) -> Iterator[List[int]]:
        yield from self.batches()
