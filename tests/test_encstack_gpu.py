"""GPU suite: the text-encoder forward as one persistent launch (kk_encoder_stack_fwd, csrc/kk_encstack.hip) against the
per-kernel sequence it replaces — every tensor the backward reads, for several batch shapes, both workgroup placements
(groups inside an XCD / groups spread over all XCDs: the hand-offs must not depend on it), dropout on and off — and under
load: hundreds of back-to-back graph replays beside the other branches of the step must give the same bits every time."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SAVED = ["sa.qkv_raw", "sa.qkv_n", "sa.ctx", "sa.lse", "xm", "ln2.y", "ln2.mean", "ln2.rstd", "ff.h1", "ff.g", "ff.f2", "ff.rstd_f", "xo"]


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from kokoro_ruslan_amd import engine, spec, synthetic
    return engine, spec, synthetic


def _engine(mods, dims=None):
    engine, spec, _ = mods
    e = engine.KokoroEngine(dims or spec.ModelDims(), spec.StepHyper(gradient_accumulation_steps=1), math_mode="bf16",
                            total_steps=20000, seed=0)
    return e


def _encoder_tensors(e, B, P):
    """Clones of everything the encoder forward leaves behind for the backward (views of the workspace)."""
    d, out = e.dims, {}
    Ne, H, F, h = B * P, e.dims.hidden, e.dims.enc_ff, e.dims.heads
    shapes = {"sa.qkv_raw": (Ne, 3 * H), "sa.qkv_n": (Ne, 3 * H), "sa.ctx": (Ne, H), "sa.lse": (B, h, P), "xm": (Ne, H), "ln2.y": (Ne, H),
              "ln2.mean": (Ne,), "ln2.rstd": (Ne,), "ff.h1": (Ne, 2 * F), "ff.g": (Ne, F), "ff.f2": (Ne, H), "ff.rstd_f": (Ne,), "xo": (Ne, H)}
    f32 = {"sa.lse", "xm", "ln2.mean", "ln2.rstd", "ff.rstd_f", "xo"}
    for i in range(d.enc_layers):
        for k in SAVED:
            out[f"enc{i}.{k}"] = e._buf(f"enc{i}.{k}", *shapes[k], dtype=torch.float32 if k in f32 else e.enc_dt).clone()
        if i > 0:
            for k, shp, dt in ((".ln1.y", (Ne, H), e.enc_dt), (".ln1.mean", (Ne,), torch.float32), (".ln1.rstd", (Ne,), torch.float32)):
                out[f"enc{i}{k}"] = e._buf(f"enc{i}{k}", *shp, dtype=dt).clone()
    out["enc.norm.y"] = e._buf("enc.norm.y", Ne, H).clone()
    out["enc.norm.mean"], out["enc.norm.rstd"] = e._buf("enc.norm.mean", Ne).clone(), e._buf("enc.norm.rstd", Ne).clone()
    return out


def _forward(e, batch, fused, placement=0, seed=77):
    e.enc_fused, e.enc_placement = fused, placement
    e.rng.fill_(seed)
    out = e.forward_backward(batch, backward=False)
    torch.cuda.synchronize()
    B, P = batch["phoneme_indices"].shape
    return _encoder_tensors(e, B, P), out["losses"].clone()


def _rel(a, b):
    a, b = a.double(), b.double()
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin)
    return float((a[fin] - b[fin]).norm() / (b[fin].norm() + 1e-30))


@pytest.mark.parametrize("B,T,P", [(8, 512, 64), (8, 256, 128), (8, 1024, 128), (3, 96, 50), (16, 128, 64), (2, 64, 33), (5, 160, 97)])
@pytest.mark.parametrize("dropout", [True, False])
def test_fused_encoder_matches_the_per_kernel_sequence(mods, B, T, P, dropout):
    _, _, synthetic = mods
    e = _engine(mods)
    assert e._encoder_stack_ok(B, P) == (B <= 8), "the engine takes the fused launch for up to 8 items (one per workgroup group)"
    e.enc_fused_max_batch = 64                           # (the launch itself serves any batch: a group walks its items in turn)
    assert e._encoder_stack_ok(B, P), "the shape must be served by the fused launch"
    e.train_dropout = dropout
    batch = {k: v.cuda() for k, v in synthetic.synthetic_batch(B, T, P, seed=5, ragged=B in (3, 5, 16)).items()}     # ragged: padded keys
    ref, ref_loss = _forward(e, batch, fused=False)
    for placement in (0, 1):
        got, loss = _forward(e, batch, fused=True, placement=placement)
        assert e.encoder_stack_error() == 0, "a group barrier of the fused encoder timed out"
        worst = []
        for name, r in ref.items():
            depth = e.dims.enc_layers if name.startswith("enc.norm") else int(name[3]) + 1
            err = _rel(got[name], r)
            # same arithmetic up to summation order inside a dot product; a flipped bf16 rounding is amplified layer by layer
            assert err < 6e-3 * depth, f"placement {placement}: {name} rel. error {err:.3e}"
            worst.append((err, name))
        # the dropout masks are the same functions of (seed, site, element): zeros in the same places
        g_ref, g_got = ref["enc0.ff.g"], got["enc0.ff.g"]
        same = float(((g_ref == 0) == (g_got == 0)).float().mean())
        assert same > 0.9999, f"GLU dropout masks differ ({same})"
        assert abs(float(loss[0]) - float(ref_loss[0])) < 2e-2 * abs(float(ref_loss[0])), (loss, ref_loss, max(worst))


def test_fused_encoder_first_layer_is_tight(mods):
    """Layer 0 sees identical inputs on both paths: differences are single bf16 roundings."""
    _, _, synthetic = mods
    e = _engine(mods)
    e.train_dropout = True
    batch = {k: v.cuda() for k, v in synthetic.synthetic_batch(8, 512, 64, seed=9).items()}
    ref, _ = _forward(e, batch, fused=False)
    got, _ = _forward(e, batch, fused=True)
    for k in ("sa.qkv_raw", "sa.qkv_n", "sa.ctx", "sa.lse", "xm", "ln2.y", "ff.h1", "ff.g", "ff.f2", "xo"):
        err = _rel(got["enc0." + k], ref["enc0." + k])
        assert err < 5e-3, f"enc0.{k}: {err:.3e}"
        if k == "sa.qkv_raw":      # the summation order differs, the values do not: almost all elements bit-equal
            eq = float((got["enc0." + k] == ref["enc0." + k]).float().mean())
            assert eq > 0.97, eq


def test_fused_encoder_is_deterministic_under_load(mods):
    """300 replays of the whole captured step (the decoder head and the gradient zero-fill run beside the encoder launch): with
    the seed pinned, the encoder's output must be bit-identical every time — a stale hand-off would show up as a difference."""
    _, _, synthetic = mods
    e = _engine(mods)
    e.train_dropout = True
    batch = {k: v.cuda() for k, v in synthetic.synthetic_batch(8, 512, 64, seed=11).items()}
    for placement in (0, 1):
        e.enc_placement = placement
        e._invalidate()
        g = None
        first = None
        for it in range(3):
            e.rng.fill_(123)
            e.forward_backward(batch, zero_grads=True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        e.rng.fill_(123)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            e.forward_backward(batch, zero_grads=True)
        bad = 0
        for it in range(300):
            e.rng.fill_(123)
            g.replay()
            if it % 10 == 0 or it > 280:
                y = e._buf("enc.norm.y", 8 * 64, e.dims.hidden).clone()
                if first is None:
                    first = y
                bad += int(not torch.equal(y, first))
        torch.cuda.synchronize()
        assert e.encoder_stack_error() == 0
        assert bad == 0, f"placement {placement}: {bad} replays differed"
        assert torch.isfinite(first).all()


def test_fused_encoder_train_steps_track_the_unfused_engine(mods):
    """A few optimizer steps with the fused encoder against the same steps without it (same seeds, dropout on)."""
    _, _, synthetic = mods
    batch = {k: v.cuda() for k, v in synthetic.synthetic_batch(8, 256, 64, seed=13).items()}
    losses = {}
    for fused in (False, True):
        e = _engine(mods)
        e.train_dropout = True
        e.enc_fused = fused
        ls = []
        for _ in range(6):
            ls.append(e.train_step_graphed(batch).clone())
        torch.cuda.synchronize()
        losses[fused] = torch.stack(ls).cpu()
        if fused:
            assert e.encoder_stack_error() == 0
            assert e.opt_stats()["skipped"] == 0
    assert torch.isfinite(losses[True]).all()
    assert torch.allclose(losses[True][:, 0], losses[False][:, 0], rtol=3e-2), (losses[True][:, 0], losses[False][:, 0])


def test_unsupported_shapes_fall_back(mods):
    e = _engine(mods)
    assert not e._encoder_stack_ok(4, 129)          # longer than a group carries
    e.enc_fused = False
    assert not e._encoder_stack_ok(8, 64)
