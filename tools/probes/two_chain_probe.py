"""Does the step's chain of thin dependent launches leave room for a SECOND, independent chain beside it?

Two engines (own arenas, workspaces, streams), each with half of the 8x512 batch, replay their captured forward+backward
concurrently from two host threads; compared with one engine replaying the whole batch.  forward+backward only (the
optimizer pass is HBM-bound and would run once either way).  GPU_MAX_HW_QUEUES decides how many streams get their own hardware queue.
(Round 2 result: 5.6-7.4 ms for the two half-batch chains against 4.46 ms for one chain of the whole batch.)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from kokoro_ruslan_amd.engine import KokoroEngine
from kokoro_ruslan_amd.spec import ModelDims, StepHyper
from kokoro_ruslan_amd.synthetic import synthetic_batch

T, P = int(os.environ.get("FRAMES", 512)), int(os.environ.get("PHONEMES", 64))
N = int(os.environ.get("STEPS", 100))


def make(B, seed):
    eng = KokoroEngine(ModelDims(), StepHyper(gradient_accumulation_steps=1), math_mode="bf16", total_steps=20000, seed=0)
    eng.train_dropout = True
    batch = {k: v.cuda() for k, v in synthetic_batch(B, T, P, seed=seed).items()}
    for _ in range(3):
        eng.train_step_graphed(batch)
    torch.cuda.synchronize()
    ent = next(iter(eng._graphs.values()))
    fb = next(iter(ent["fb"].values()))
    return eng, fb


def replay(eng, fb):
    if fb["prog"] is not None:
        eng._run_program(fb["prog"])
    else:
        fb["g1"].replay()


def timed(pairs):
    """pairs = [(engine, fb, stream)]: every pair replayed N times from its own host thread; wall time of all."""
    def work(eng, fb, st):
        with torch.cuda.stream(st):
            for _ in range(N):
                replay(eng, fb)
    for e, f, s in pairs:
        with torch.cuda.stream(s):
            replay(e, f)
    torch.cuda.synchronize()
    th = [threading.Thread(target=work, args=p) for p in pairs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


whole = make(8, 1)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
t8 = timed([(whole[0], whole[1], sa)])
print(f"one chain, 8x{T}: {t8:.3f} ms per forward+backward")
a, b = make(4, 2), make(4, 3)
t4 = timed([(a[0], a[1], sa)])
print(f"one chain, 4x{T}: {t4:.3f} ms")
t44 = timed([(a[0], a[1], sa), (b[0], b[1], sb)])
print(f"two chains of 4x{T} side by side: {t44:.3f} ms  (vs {t8:.3f} for one chain of 8: x{t8 / t44:.2f})")
c = make(8, 4)
t88 = timed([(whole[0], whole[1], sa), (c[0], c[1], sb)])
print(f"two chains of 8x{T} side by side: {t88:.3f} ms for 16 samples (vs {2 * t8:.3f} in sequence: x{2 * t8 / t88:.2f})")
