"""Build libkokoro_hip.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python -m kokoro_ruslan_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
repo snapshot (see README).  No torch headers are involved: the boundary is plain C.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libkokoro_hip.so")
SOURCES = ["kk_core.hip", "kk_gemm.hip", "kk_gemm16.hip", "kk_gemm16x.hip", "kk_attn.hip", "kk_norm.hip", "kk_elem.hip", "kk_loss.hip", "kk_optim.hip", "kk_dropout.hip", "kk_comm.hip", "kk_encstack.hip", "kk_chain.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, tuning: bool = False, variant: str = "", defs=()) -> str:
    """tuning=True builds the TOOLS flavour next to the product library (libkokoro_hip_tuning.so, objects under _obj_tuning): it
    also exports the tuning hooks of include/kokoro_hip_tuning.h and reads the KK_* A/B switches and timing probes from the
    environment (kk_common.h: kk_tune_env / KK_DBG), which the product build folds to their defaults.  Tools pick it with
    `bench.py --lib tuning` / kokoro_ruslan_amd.lib.use_library().
    variant="name", defs=["-DX=1", ...]: a tools flavour compiled with extra definitions as libkokoro_hip_<name>.so (compile-time
    A/B of a kernel change on one box: `bench.py --lib <name>`, tools/probes/ab.sh)."""
    if variant and not os.environ.get("KK_VARIANT_PRODUCT"):       # KK_VARIANT_PRODUCT=1: the PRODUCT flavour + the definitions (an A/B against the product
        tuning = True                                               # library itself: the tools flavour is ~1 % slower in the step, so never compare across flavours)
    tag = variant or ("tuning" if tuning else "")
    obj_dir = OBJ + (f"_{tag}" if tag else "")
    lib_path = LIB.replace(".so", f"_{tag}.so") if tag else LIB
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    flags = FLAGS + (["-DKK_TUNING_HOOKS"] if tuning else []) + list(defs)
    # the objects of a flavour are only as good as the flags they were built with: a stamp of the flags forces a rebuild when the
    # same --variant name comes back with other -D definitions (an A/B could otherwise compare identical code)
    # (the stamp is removed first and written only after every object and the link succeeded: a failed or interrupted build leaves
    #  no stamp, so the next one rebuilds everything instead of linking old-flag objects beside new-flag ones — ADVICE r4)
    stamp = os.path.join(obj_dir, "flags.stamp")
    want = " ".join(flags)
    stamp_ok = os.path.exists(stamp) and open(stamp).read() == want
    if not stamp_ok:
        force = True
        if os.path.exists(stamp):
            os.remove(stamp)
    headers = [os.path.join(CSRC, "kk_common.h"), os.path.join(os.path.dirname(HERE), "include", "kokoro_hip.h")]
    headers += [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".inc") or (f.endswith(".h") and f != "kk_common.h")]      # kernel bodies included by kk_attn.hip
    chain_deps = [os.path.join(CSRC, f) for f in ("kk_gemm16x.hip", "kk_gemm16.hip", "kk_attn.hip", "kk_dropout.hip")]      # bodies compiled into kk_chain.hip

    def compile_one(src: str) -> str:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers + (chain_deps if src == "kk_chain.hip" else [])):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
            if verbose and r.stderr.strip():
                print(r.stderr, file=sys.stderr)
        return o

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(lib_path, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs + ["-ldl"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    if not stamp_ok:
        with open(stamp, "w") as f:
            f.write(want)
    return lib_path


if __name__ == "__main__":
    av = sys.argv[1:]
    var = av[av.index("--variant") + 1] if "--variant" in av else ""
    print(build(force="--force" in av, verbose=True, tuning="--tuning" in av, variant=var, defs=[a for a in av if a.startswith("-D")]))
