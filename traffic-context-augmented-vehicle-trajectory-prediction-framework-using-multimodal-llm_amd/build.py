"""In-tree build of the gfx950 C-ABI library (csrc/*.hip -> libtcavt_hip.so).

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the
repo snapshot.  Usage: ``python -m tcavt_amd.build`` or ``build_library()``.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libtcavt_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ["core.hip", "gemm_bf16.hip", "attention.hip", "small.hip", "backward.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-std=c++17", "-Wall"]


def _digest():
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    for n in names:
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(n.encode())
            h.update(f.read())
    with open(os.path.join(HERE, "..", "include", "tcavt.h"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_current():
    stamp = LIB_PATH + ".sha256"
    if not (os.path.exists(LIB_PATH) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        return f.read().strip() == _digest()


EXP_LIB_PATH = os.path.join(HERE, "libtcavt_hip_exp.so")


def build_library(force=False, verbose=True, experiments=False):
    """Compile every HIP source for gfx950 and link the shared library.

    experiments=True builds libtcavt_hip_exp.so with -DTCAVT_EXPERIMENTS instead: the product library plus the
    measured-and-rejected GEMM variants and the timing-only elimination experiments (which compute wrong results).
    Only tools/ load it (TCAVT_LIB=exp); tests, bench.py and smoke() never do."""
    if experiments:
        return _build(EXP_LIB_PATH, FLAGS + ["-DTCAVT_EXPERIMENTS"], "build_exp", verbose, stamp=False)
    if not force and is_current():
        if verbose:
            print(f"[tcavt build] {LIB_NAME} up to date")
        return LIB_PATH
    return _build(LIB_PATH, FLAGS, "build", verbose, stamp=True)


def _build(lib_path, flags, objsub, verbose, stamp):
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    sources += sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f not in sources)
    objdir = os.path.join(HERE, objsub)
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, sources))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    if stamp:
        with open(lib_path + ".sha256", "w") as f:
            f.write(_digest())
    if verbose:
        print(f"[tcavt build] built {lib_path}")
    return lib_path


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, experiments="--experiments" in sys.argv)
