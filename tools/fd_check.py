"""Diagnostic: directional-derivative check per parameter tensor, with and without dropout (same masks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import kokoro_oracle as O
from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper

fx = np.load("tests/golden/tiny_ragged.npz")
d = O.ModelDims(*[int(x) for x in fx["dims"]])
batch = {k.split("/", 1)[1]: torch.from_numpy(fx[k]).cuda() for k in fx.files if k.startswith("batch/")}
P = O.init_params(d, int(fx["seed"]))
names = list(P)
which = dict(sa=(0.2, 0, 0, 0), all=None)
for mode in ("off", "on", "attn_only", "resid_only", "glu_only", "droppath_only", "var_only", "specaug_only", "input_only"):
    hp = StepHyper()
    if mode == "attn_only": pass
    e = KokoroEngine(ModelDims(**d.__dict__), hp, init=False)
    e.load_params(P)
    e.train_dropout = mode != "off"
    # monkeypatch selective sites
    if mode not in ("off", "on"):
        orig_resid, orig_resid_b = e._residual, e._residual_bwd
        import kokoro_ruslan_amd.lib as kk
        real_call = kk.call
        def call(name, *a, _mode=mode):
            a = list(a)
            def zero_p(i):
                a[i] = 0.0
            if name in ("kk_attn_fwd", "kk_attn_bwd_dq", "kk_attn_bwd_dkv") and _mode != "attn_only": zero_p(-3)
            if name in ("kk_glu_fwd", "kk_glu_bwd") and _mode != "glu_only": zero_p(-2)
            if name in ("kk_embed_fwd", "kk_embed_bwd") and _mode != "input_only": zero_p(-1)
            if name == "kk_groupnorm_relu_fwd" and _mode != "var_only": zero_p(-1)
            if name == "kk_groupnorm_relu_bwd" and _mode != "var_only": zero_p(-1)
            if name == "kk_specaug" and _mode != "specaug_only": return
            if name in ("kk_dropout_fwd", "kk_dropout_bwd"):
                # (… seed, site1, p1, site2, p2, site_dp, dp_rate[, dx_bf16])
                tail = [a.pop()] if name == "kk_dropout_bwd" else []
                if _mode == "resid_only": a[-1] = 0.0
                elif _mode == "droppath_only": a[-5] = 0.0; a[-3] = 0.0
                elif _mode == "input_only":
                    if a[-6] not in (30, 31): a[-5] = 0.0; a[-3] = 0.0; a[-1] = 0.0
                else: a[-5] = 0.0; a[-3] = 0.0; a[-1] = 0.0
                a += tail
            return real_call(name, *a)
        import kokoro_ruslan_amd.engine as em
        em.kk.call = call
    def run(params, backward):
        e.load_params(params, reset_ema=False); e.rng.fill_(100); e.zero_grad()
        out = e.forward_backward(batch, backward=backward); torch.cuda.synchronize()
        return float(out["losses"][0]), {n: g.clone().cpu() for n, g in e.grads().items()}
    l, g = run(P, True)
    gen = torch.Generator().manual_seed(0)
    res = []
    for key in ("decoder.layers.0.ff.linear1.weight", "decoder.layers.1.cross_attn.w_k.weight", "decoder.layers.0.self_attn.w_q.weight",
                "transformer_encoder_layers.0.self_attn.w_q.weight", "pitch_predictor.conv_layers.0.weight", "mel_projection_in.weight",
                "pitch_embedding.weight", "duration_predictor.conv_layers.0.weight", "text_embedding.weight"):
        n = [x for x in names if key in x][0]
        V = torch.randn(P[n].shape, generator=gen)
        an = float((g[n].double() * V.double()).sum())
        eps = 1e-3
        lp, _ = run({**P, n: P[n] + eps * V}, False)
        lm, _ = run({**P, n: P[n] - eps * V}, False)
        fd = (lp - lm) / (2 * eps)
        res.append(f"{key.split('.')[-3] if 'layers' in key else key.split('.')[0]}:{an:+.4f}/{fd:+.4f}")
    print(f"[{mode:14s}] loss {l:.5f} | analytic/fd: " + "  ".join(res))
    if mode not in ("off", "on"):
        em.kk.call = real_call
