"""SURVEY §8(f)4 — autoregressive decoding.  CPU: the oracle's restatement of forward_inference against the fixture dumped
from the reference (tests/golden/make_golden.py: section_inference).  GPU: KokoroEngine.generate against the same fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import kokoro_oracle as O

CASES = ("never_stops", "stops_at_min_length", "batch2_padded_stress")


def _fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, "inference_tiny.npz"))
    d = O.ModelDims(*[int(x) for x in fx["dims"]])
    seed = int(fx["seed"])
    P = O.init_params(d, seed)
    g = torch.Generator().manual_seed(seed + 1)
    for n, p in P.items():
        if p.dim() == 1:
            p.add_(torch.randn(p.shape, generator=g) * 0.1)
    return fx, d, P


def _case(fx, name):
    ids = torch.from_numpy(fx[f"{name}/ids"])
    stress = torch.from_numpy(fx[f"{name}/stress"]) if f"{name}/stress" in fx.files else None
    max_len, thr = fx[f"{name}/kw"]
    return ids, stress, dict(max_len=int(max_len), stop_threshold=float(thr)), torch.from_numpy(fx[f"{name}/mel"])


@pytest.fixture(scope="module")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", CASES)
def test_oracle_generate_matches_the_reference(golden_dir, name):
    fx, d, P = _fixture(golden_dir)
    ids, stress, kw, ref = _case(fx, name)
    mel, info = O.generate(P, O.make_buffers(d), ids, stress, d, want=True, **kw)
    assert mel.shape == ref.shape, "same number of frames: same stop decision"
    assert torch.equal(info["durations"], torch.from_numpy(fx[f"{name}/durations"]))
    torch.testing.assert_close(mel, ref, atol=2e-5, rtol=0)


def test_oracle_stop_rules():
    """The stop head may fire only from min_expected_length on, and the firing frame is kept."""
    d = O.ModelDims(vocab=59, mel=20, hidden=128, heads=2, enc_layers=1, dec_layers=1, enc_ff=96, dec_ff=96, var_filter=32,
                    var_kernel=3, var_bins=16, max_len=700)
    P, Bf = O.init_params(d, 3), O.make_buffers(d)
    ids = torch.randint(1, 59, (1, 5), generator=torch.Generator().manual_seed(1))
    mel, info = O.generate(P, Bf, ids, None, d, max_len=200, stop_threshold=0.0, want=True)
    lo, expected, hi = info["bounds"]
    assert mel.shape[1] == lo + 1 and lo == max(12, int(expected * 0.7))
    mel2 = O.generate(P, Bf, ids, None, d, max_len=lo + 5, stop_threshold=2.0)
    assert mel2.shape[1] == lo + 5, "no stop: runs to the bound"
    assert float(mel2.max()) <= 2.0 and float(mel2.min()) >= -11.5


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("check_every,decode_graph", [(1, False), (16, False), (1, True), (16, True)])
def test_engine_generate_matches_the_reference(golden_dir, name, check_every, decode_graph):
    from kokoro_ruslan_amd import engine as eng_mod
    from kokoro_ruslan_amd.spec import ModelDims, StepHyper
    fx, d, P = _fixture(golden_dir)
    ids, stress, kw, ref = _case(fx, name)
    e = eng_mod.KokoroEngine(ModelDims(**d.__dict__), StepHyper(), math_mode="f32", init=False, total_steps=100)
    e.load_params(P)
    mel = e.generate(ids.cuda(), stress.cuda() if stress is not None else None, check_every=check_every, decode_graph=decode_graph, **kw)
    assert tuple(mel.shape) == tuple(ref.shape), "same number of frames: same stop decision"
    torch.testing.assert_close(mel.cpu(), ref, atol=1e-4, rtol=0)          # the mel-L1 bar of the train step (fp32 mode)
    e.train_dropout = True
    e.generate(ids.cuda(), stress.cuda() if stress is not None else None, **kw)
    assert e.train_dropout is True, "generate() runs with dropout off and restores the flag"


@pytest.mark.gpu
def test_engine_generate_bf16_mode(golden_dir):
    """bf16 storage + fused epilogue kernels at Sq = 1: same frame count when the stop head is disabled, close values."""
    from kokoro_ruslan_amd import engine as eng_mod
    from kokoro_ruslan_amd.spec import ModelDims, StepHyper
    fx, d, P = _fixture(golden_dir)
    ids, stress, kw, ref = _case(fx, "batch2_padded_stress")
    e = eng_mod.KokoroEngine(ModelDims(**d.__dict__), StepHyper(), math_mode="bf16", init=False, total_steps=100)
    e.load_params(P)
    mel = e.generate(ids.cuda(), stress.cuda(), **kw).cpu()
    assert tuple(mel.shape) == tuple(ref.shape) and bool(torch.isfinite(mel).all())
    assert float((mel - ref).abs().mean()) < 0.08 * float(ref.abs().mean()) + 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_engine_generate_graph_replay_equals_eager_launches(golden_dir, mode):
    """The decoder step replayed from ONE hipGraph (its launches index the frame by a device-side counter: kk_decode_prologue /
    kk_decode_cache_append / kk_decode_epilogue) gives what the same launches give when issued one by one — default model size, 90
    frames, batch of 2 with padding.  (Not bit for bit: one-row GEMMs slice their reduction and add the slices with fp32 atomics, whose
    order differs from run to run; the error does not grow over the 90 autoregressive steps.)"""
    from kokoro_ruslan_amd import engine as eng_mod
    from kokoro_ruslan_amd.spec import ModelDims, StepHyper
    e = eng_mod.KokoroEngine(ModelDims(), StepHyper(), math_mode=mode, total_steps=100, seed=3)
    ids = torch.randint(1, 59, (2, 40), generator=torch.Generator().manual_seed(1))
    ids[1, 33:] = 0
    kw = dict(max_len=90, stop_threshold=2.0, min_len_ratio=0.0, min_len_floor=1)
    a = e.generate(ids.cuda(), decode_graph=False, **kw)
    b = e.generate(ids.cuda(), decode_graph=True, check_every=7, **kw)
    c = e.generate(ids.cuda(), decode_graph=True, **kw)
    assert a.shape[1] >= 30 and bool(torch.isfinite(a).all())
    tol = 1e-3 if mode == "f32" else 0.05
    for other in (b, c):
        assert other.shape == a.shape and float((other - a).abs().max()) <= tol * max(1.0, float(a.abs().max()))
