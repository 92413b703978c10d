"""`kokoro-train` entry point (reference cli/training.py:19-75)."""
from __future__ import annotations

import logging
import sys

from kokoro.cli.cli import create_config_from_args, parse_arguments


def main(argv=None) -> int:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    args = parse_arguments(argv)
    config = create_config_from_args(args)
    from kokoro.training.trainer import KokoroTrainer
    trainer = KokoroTrainer(config)
    trainer.train()
    return 0


if __name__ == "__main__":
    sys.exit(main())
