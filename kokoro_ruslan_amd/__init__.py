"""kokoro_ruslan_amd — MI355X-native engine for the Kokoro acoustic-model train step.

Only what the hot path needs: ``csrc/`` (HIP kernels + C ABI), ``lib`` (ctypes binding),
``engine`` (host step driver: flat parameter arena, explicit forward/backward kernel sequences,
fused optimizer pass), ``dp`` (data-parallel gradient exchange over RCCL).
"""
__version__ = "0.1.0"
