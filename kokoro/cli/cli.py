"""`kokoro-train` command line: the reference's flags, defaults and dest names (reference cli/cli.py:33-220) and
the flag → TrainingConfig mapping (cli/cli.py:226-292)."""
from __future__ import annotations

import argparse

import torch

from kokoro.training.config import TrainingConfig

# (flags, kwargs) table — one row per reference flag
_FLAGS = [
    (("--corpus", "-c"), dict(type=str, default="./ruslan_corpus", help="corpus directory")),
    (("--output", "-o"), dict(type=str, default="./kokoro_russian_model", help="output model directory")),
    (("--resume", "-r"), dict(type=str, default=None, help='"auto" or a checkpoint path')),
    (("--batch-size", "-b"), dict(type=int, default=8)),
    (("--epochs", "-e"), dict(type=int, default=None)),
    (("--learning-rate", "-lr"), dict(type=float, default=None)),
    (("--save-every",), dict(type=int, default=5)),
    (("--mfa-alignments",), dict(type=str, default=None)),
    (("--no-mfa",), dict(action="store_true")),
    (("--val-split",), dict(type=float, default=0.1)),
    (("--no-validation",), dict(action="store_true")),
    (("--early-stopping-patience",), dict(type=int, default=10)),
    (("--validation-interval",), dict(type=int, default=1)),
    (("--dynamic-batching",), dict(action="store_true", default=True)),
    (("--no-dynamic-batching",), dict(action="store_false", dest="dynamic_batching")),
    (("--max-frames",), dict(type=int, default=None)),
    (("--min-batch-size",), dict(type=int, default=4)),
    (("--max-batch-size",), dict(type=int, default=32)),
    (("--profile-amp",), dict(action="store_true")),
    (("--profile-amp-batches",), dict(type=int, default=10)),
    (("--fused-adamw",), dict(action="store_true")),
    (("--no-fused-adamw",), dict(action="store_true")),
    (("--try-fused-adamw-mps",), dict(action="store_true", default=True)),
    (("--verbose", "-v"), dict(action="store_true")),
    (("--no-memory-cache",), dict(action="store_false", dest="use_memory_cache")),
    (("--stop-threshold",), dict(dest="stop_threshold", type=float, default=0.1)),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Kokoro acoustic-model training (MI355X engine)")
    for flags, kw in _FLAGS:
        p.add_argument(*flags, **kw)
    amp = p.add_mutually_exclusive_group()
    amp.add_argument("--enable-amp", action="store_true")
    amp.add_argument("--disable-amp", action="store_true")
    return p


def parse_arguments(argv=None):
    return build_parser().parse_args(argv)


def create_config_from_args(args) -> TrainingConfig:
    use_amp = True if args.enable_amp else False if args.disable_amp else torch.cuda.is_available()
    kw = dict(
        data_dir=args.corpus, output_dir=args.output, batch_size=args.batch_size, sample_rate=22050, hop_length=256,
        win_length=1024, n_fft=1024, n_mels=80, f_min=0.0, f_max=8000.0, save_every=args.save_every,
        use_mixed_precision=use_amp, use_dynamic_batching=args.dynamic_batching,
        max_frames_per_batch=args.max_frames if args.max_frames is not None else 30000,
        min_batch_size=args.min_batch_size, max_batch_size=args.max_batch_size, use_mfa=not args.no_mfa,
        mfa_alignment_dir=args.mfa_alignments or "./mfa_output/alignments", num_workers=0, pin_memory=False,
        resume_checkpoint=args.resume, validation_split=0.0 if args.no_validation else args.val_split,
        validation_interval=args.validation_interval, early_stopping_patience=args.early_stopping_patience,
        use_fused_adamw=(False if args.no_fused_adamw else True if args.fused_adamw else None),
        try_fused_adamw_on_mps=args.try_fused_adamw_mps, verbose=args.verbose,
        use_memory_cache=getattr(args, "use_memory_cache", True))
    if args.learning_rate is not None:
        kw["learning_rate"] = args.learning_rate
    if args.epochs is not None:
        kw["num_epochs"] = args.epochs
    return TrainingConfig(**kw)
