"""Seeded synthetic padded batches with the reference's collate contract (data/dataset.py:871-921), shapes and
value ranges of SURVEY §8d: phoneme ids U[1,V), stress p(1)=0.15, durations = even split (dataset.py:581-606) with a
±2 jitter re-normalised to Σ = mel length, log-mel N(-5, 2²) clamped to [-11.5, 2], pitch U[0,1] with 30 % zeros,
energy U[0,1], smoothed stop targets (dataset.py:32-64: tail 6, decay 0.5)."""
from __future__ import annotations

from typing import Dict

import torch


def stop_targets(T: int, tail: int = 6, decay: float = 0.5) -> torch.Tensor:
    t = torch.zeros(T)
    if T > 0:
        n = min(tail + 1, T)
        t[T - n:T] = (decay ** torch.arange(n, dtype=torch.float32)).flip(0)
    return t


def synthetic_batch(B: int, T: int, P: int, vocab: int = 59, mel: int = 80, seed: int = 1234,
                    ragged: bool = False, lengths=None) -> Dict[str, torch.Tensor]:
    """lengths = (mel lengths [B], phoneme lengths [B]): a batch padded to T x P with those valid lengths (what collate_fn makes of
    B utterances); ragged=True draws them in [0.6, 1] x (T, P) with sample 0 full length."""
    g = torch.Generator().manual_seed(seed)
    mel_len = torch.full((B,), T, dtype=torch.long)
    ph_len = torch.full((B,), P, dtype=torch.long)
    if lengths is not None:
        mel_len, ph_len = torch.as_tensor(lengths[0], dtype=torch.long), torch.as_tensor(lengths[1], dtype=torch.long)
        assert mel_len.shape == (B,) and ph_len.shape == (B,) and int(mel_len.max()) <= T and int(ph_len.max()) <= P
    elif ragged and B > 1:
        mel_len[1:] = (T * (0.6 + 0.4 * torch.rand(B - 1, generator=g))).long().clamp(min=4)
        ph_len[1:] = (P * (0.6 + 0.4 * torch.rand(B - 1, generator=g))).long().clamp(min=2)
    out = {"phoneme_indices": torch.zeros(B, P, dtype=torch.long), "stress_indices": torch.zeros(B, P, dtype=torch.long),
           "phoneme_durations": torch.zeros(B, P, dtype=torch.long), "mel_specs": torch.zeros(B, T, mel),
           "pitches": torch.zeros(B, T), "energies": torch.zeros(B, T), "stop_token_targets": torch.zeros(B, T),
           "mel_lengths": mel_len, "phoneme_lengths": ph_len}
    for b in range(B):
        t, p = int(mel_len[b]), int(ph_len[b])
        out["phoneme_indices"][b, :p] = torch.randint(1, vocab, (p,), generator=g)
        out["stress_indices"][b, :p] = (torch.rand(p, generator=g) < 0.15).long()
        base = torch.full((p,), t // p, dtype=torch.long)
        base[: t % p] += 1
        dd = (base + torch.randint(-2, 3, (p,), generator=g)).clamp(min=1)
        diff, k = t - int(dd.sum()), 0
        while diff != 0:
            j = k % p
            if diff > 0:
                dd[j] += 1
                diff -= 1
            elif dd[j] > 1:
                dd[j] -= 1
                diff += 1
            k += 1
        out["phoneme_durations"][b, :p] = dd
        out["mel_specs"][b, :t] = (torch.randn(t, mel, generator=g) * 2 - 5).clamp(-11.5, 2.0)
        pv = torch.rand(t, generator=g)
        pv[torch.rand(t, generator=g) < 0.3] = 0.0
        out["pitches"][b, :t] = pv
        out["energies"][b, :t] = torch.rand(t, generator=g)
        out["stop_token_targets"][b, :t] = stop_targets(t)
    return out
