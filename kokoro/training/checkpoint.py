"""Checkpoint files in the reference's layout, so its inference / vocoder tooling loads what this engine trains.

Reference: save_checkpoint_with_scaler (training/trainer.py:1987-2036), build_model_metadata / find_latest_checkpoint /
save_final_model (training/checkpoint_manager.py:178-241, 898-925).  File names `checkpoint_epoch_{N}.pth`,
`kokoro_russian_final.pth`; state-dict keys are the reference's un-prefixed parameter names; `optimizer_state_dict`
has torch.optim.AdamW's structure with the reference's 10 param groups (each tagged `group_type`).
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Any, Dict, Optional

import torch

from kokoro_ruslan_amd import lib as kk
from kokoro_ruslan_amd import spec


def build_model_metadata(config, dims: spec.ModelDims) -> Dict[str, Any]:
    g = lambda k, d: getattr(config, k, d)
    return {
        "schema_version": 2,
        "architecture": {
            "mel_dim": int(dims.mel), "hidden_dim": int(dims.hidden), "n_encoder_layers": int(dims.enc_layers),
            "n_decoder_layers": int(dims.dec_layers), "n_heads": int(dims.heads), "encoder_ff_dim": int(dims.enc_ff),
            "decoder_ff_dim": int(dims.dec_ff), "encoder_dropout": float(g("encoder_dropout", 0.1)),
            "max_decoder_seq_len": int(dims.max_len), "use_variance_predictor": True,
            "variance_filter_size": int(dims.var_filter), "variance_kernel_size": int(dims.var_kernel),
            "variance_dropout": float(g("variance_dropout", 0.1)), "n_variance_bins": int(dims.var_bins),
            "pitch_min": float(g("pitch_min", 0.0)), "pitch_max": float(g("pitch_max", 1.0)),
            "energy_min": float(g("energy_min", 0.0)), "energy_max": float(g("energy_max", 1.0)),
            "use_stochastic_depth": bool(g("use_stochastic_depth", True)),
            "stochastic_depth_rate": float(g("stochastic_depth_rate", 0.1)), "qk_norm": True, "ffn_output_norm": True,
            "vocab_size": int(dims.vocab),
        },
        "inference_controls": {
            "max_len": int(g("inference_max_len", 1200)), "stop_threshold": float(g("inference_stop_threshold", 0.45)),
            "min_len_ratio": float(g("inference_min_len_ratio", 0.7)), "min_len_floor": int(g("inference_min_len_floor", 12)),
        },
    }


def adamw_state_dict(param_names, exp_avg, exp_avg_sq, steps: float, hp, last_base_lr: Optional[float] = None) -> Dict[str, Any]:
    """torch.optim.AdamW-shaped state for the reference's 10 param groups (trainer.py:503-642): params numbered
    consecutively through the groups; exp_avg / exp_avg_sq map a parameter name to its CPU moment tensor."""
    table = spec.group_lr_mult_wd(hp)
    groups = [[] for _ in range(10)]
    for n in param_names:
        groups[spec.param_group_of(n)].append(n)
    state, pgs, idx = {}, [], 0
    for gi, names in enumerate(groups):
        ids = []
        for n in names:
            if steps > 0:
                state[idx] = {"step": torch.tensor(float(steps)), "exp_avg": exp_avg(n), "exp_avg_sq": exp_avg_sq(n)}
            ids.append(idx)
            idx += 1
        base = last_base_lr if (steps > 0 and last_base_lr is not None) else hp.learning_rate
        pgs.append({"lr": base * table[gi][0], "betas": tuple(hp.adam_betas), "eps": hp.adam_eps, "weight_decay": table[gi][1],
                    "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                    "fused": True, "group_type": spec.GROUP_TYPES[gi], "params": ids})
    return {"state": state, "param_groups": pgs}


def optimizer_state_dict(engine) -> Dict[str, Any]:
    a = engine.arena
    st = engine.opt_stats()
    return adamw_state_dict(a.param_names, lambda n: a.view(a.m, n).detach().cpu().clone(),
                            lambda n: a.view(a.v, n).detach().cpu().clone(), float(st["attempt"] - st["skipped"]), engine.hp,
                            st["last_base_lr"])


def load_optimizer_state_dict(engine, osd: Dict[str, Any]) -> None:
    a = engine.arena
    groups = [[] for _ in range(10)]
    for n in a.param_names:
        groups[spec.param_group_of(n)].append(n)
    order = [n for g in groups for n in g]
    steps = 0.0
    for idx, n in enumerate(order):
        s = osd["state"].get(idx)
        if s is None:
            continue
        a.view(a.m, n).copy_(s["exp_avg"])
        a.view(a.v, n).copy_(s["exp_avg_sq"])
        steps = max(steps, float(s["step"]))
    engine.opt_state[kk.OS["ATTEMPT"]] = steps
    engine.opt_state[kk.OS["SKIPPED"]] = 0.0


def assemble_checkpoint(*, model_sd: Dict[str, torch.Tensor], ema_sd: Optional[Dict[str, torch.Tensor]], optimizer_sd: Dict[str, Any],
                        hp, dims: spec.ModelDims, config, total_steps: int, epoch: int, loss: float, steps_done: int,
                        val: Optional[Dict[str, float]] = None, best_val_loss: Optional[float] = None, best_val_epoch: int = -1,
                        extra: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """The checkpoint dictionary in the reference's layout (save_checkpoint_with_scaler, trainer.py:1987-2036) from plain
    CPU state — no engine, no GPU: what save_checkpoint writes, and what tests/golden/checkpoint_roundtrip.py feeds to the
    reference's own loaders."""
    val = val or {}
    c = spec.lr_schedule_consts(hp, total_steps)
    last = max(0, steps_done - c["warmup_steps"])
    ckpt = {
        "epoch": epoch, "global_step": steps_done,
        "model_state_dict": model_sd,
        "optimizer_state_dict": optimizer_sd,
        # OneCycleLR.load_state_dict is a plain __dict__.update: the fields below are the ones it steps from
        "scheduler_state_dict": {"last_epoch": last, "_step_count": last + 1, "total_steps": c["onecycle_steps"],
                                 "_last_lr": [g["lr"] for g in optimizer_sd["param_groups"]]},
        "current_optimizer_step": steps_done, "optimizer_steps_completed": steps_done,
        "loss": loss, "train_loss": loss, "val_loss": val.get("total"), "val_mel_loss": val.get("mel"),
        "val_stop_loss": val.get("stop"), "val_dur_loss": val.get("dur"),
        "best_val_loss": best_val_loss if best_val_loss is not None else val.get("total"), "best_val_epoch": best_val_epoch,
        "config": config, "model_metadata": build_model_metadata(config, dims),
        "scheduler_config": {"onecycle_steps": c["onecycle_steps"], "max_lr": c["max_lr"], "pct_start": c["pct_start"],
                             "div_factor": c["div_factor"], "warmup_steps": c["warmup_steps"]},
    }
    if not getattr(hp, "use_onecycle_lr", True):
        # legacy schedule (trainer.py:789-799): CosineAnnealingWarmRestarts stepped once per epoch, BEFORE the epoch's checkpoint is
        # written (trainer.py:2885-2887) — the saved state is the one after epoch + 1 steps, and the groups carry the next epoch's lr
        t_cur, t_i = spec.cosine_restart_position(epoch + 1, hp.lr_T_0, hp.lr_T_mult)
        f = spec.cosine_restart_factor(epoch + 1, hp.lr_T_0, hp.lr_T_mult)
        table = spec.group_lr_mult_wd(hp)
        for gi, g in enumerate(optimizer_sd["param_groups"]):
            g["initial_lr"] = hp.learning_rate * table[gi][0]
            g["lr"] = hp.lr_eta_min + (g["initial_lr"] - hp.lr_eta_min) * f
        ckpt["scheduler_state_dict"] = {"T_0": int(hp.lr_T_0), "T_i": t_i, "T_mult": int(hp.lr_T_mult), "eta_min": float(hp.lr_eta_min),
                                        "T_cur": t_cur, "base_lrs": [g["initial_lr"] for g in optimizer_sd["param_groups"]],
                                        "last_epoch": epoch + 1, "_step_count": epoch + 2,
                                        "_last_lr": [g["lr"] for g in optimizer_sd["param_groups"]]}
        ckpt["scheduler_config"] = {"onecycle_steps": None, "max_lr": None, "pct_start": None, "div_factor": None, "warmup_steps": 0}   # trainer.py:2016-2022
    if ema_sd is not None:
        ckpt["ema_model_state_dict"] = ema_sd
        ckpt["ema_updates"] = steps_done
    ckpt.update(extra or {})
    return ckpt


def save_checkpoint(engine, config, epoch: int, loss: float, output_dir: str, val: Optional[Dict[str, float]] = None,
                    best_val_loss: Optional[float] = None, best_val_epoch: int = -1, early_stopping_counter: int = 0) -> str:
    engine.check_encoder_stack()                   # never write weights trained through a timed-out encoder launch
    st = engine.opt_stats()
    done = int(st["attempt"] - st["skipped"])
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}
    ckpt = assemble_checkpoint(
        model_sd=cpu(engine.state_dict()), ema_sd=cpu(engine.state_dict(ema=True)) if engine.arena.ema is not None else None,
        optimizer_sd=optimizer_state_dict(engine), hp=engine.hp, dims=engine.dims, config=config, total_steps=engine.total_steps,
        epoch=epoch, loss=loss, steps_done=done, val=val, best_val_loss=best_val_loss, best_val_epoch=best_val_epoch,
        extra={"engine_opt_state": engine.opt_state.detach().cpu().clone(),     # new key: device-side step-driver state
               "engine_rng": int(engine.rng.item()),                            # new key: step seed of the dropout / DropPath masks
               "early_stopping_counter": int(early_stopping_counter)})          # (trainer.py:2918-3004 keeps it on the trainer)
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, f"checkpoint_epoch_{epoch + 1}.pth")
    torch.save(ckpt, path)
    return path


def find_latest_checkpoint(output_dir: str) -> Optional[str]:
    files = list(Path(output_dir).glob("checkpoint_epoch_*.pth")) if Path(output_dir).exists() else []
    if not files:
        return None
    files.sort(key=lambda p: int(p.stem.split("_")[-1]))
    return str(files[-1])


def load_checkpoint(engine, path: str) -> Dict[str, Any]:
    """Strict resume: validates the architecture metadata like the reference (checkpoint_manager.py:309-358)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    arch = (ckpt.get("model_metadata") or {}).get("architecture") or {}
    required = ["mel_dim", "hidden_dim", "n_encoder_layers", "n_decoder_layers", "max_decoder_seq_len", "use_variance_predictor"]
    missing = [k for k in required if k not in arch]
    if missing:
        raise RuntimeError(f"Checkpoint metadata is incomplete. Missing fields: {missing}.")
    d = engine.dims
    cur = {"mel_dim": d.mel, "hidden_dim": d.hidden, "n_encoder_layers": d.enc_layers, "n_decoder_layers": d.dec_layers,
           "max_decoder_seq_len": d.max_len, "use_variance_predictor": True, "vocab_size": d.vocab}
    bad = {k: (arch[k], v) for k, v in cur.items() if k in arch and arch[k] != v}
    if bad:
        raise RuntimeError(f"Checkpoint architecture mismatch (checkpoint, current): {bad}")
    engine.load_state_dict(ckpt["model_state_dict"], strict=True)
    if "ema_model_state_dict" in ckpt and engine.arena.ema is not None:
        for n, t in ckpt["ema_model_state_dict"].items():
            engine.arena.E[n].copy_(t)
    if "optimizer_state_dict" in ckpt:
        load_optimizer_state_dict(engine, ckpt["optimizer_state_dict"])
    if "engine_opt_state" in ckpt:
        engine.opt_state.copy_(ckpt["engine_opt_state"])
    if "engine_rng" in ckpt:
        engine.rng.fill_(int(ckpt["engine_rng"]))
    return ckpt


def save_final_model(engine, config, output_dir: str) -> str:
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "kokoro_russian_final.pth")
    engine.check_encoder_stack()
    torch.save({"model_state_dict": {k: v.detach().cpu().clone() for k, v in engine.state_dict().items()},
                "config": config, "model_metadata": build_model_metadata(config, engine.dims)}, path)
    return path
