"""Drop-in Python surface of the reference package for the acoustic-model train step: same module paths
(`kokoro.training.config.TrainingConfig`, `kokoro.cli.training:main`), same CLI flags, same checkpoint layout —
with the step itself executed by the MI355X engine in `kokoro_ruslan_amd`."""
