"""Build liblzzx_nerf_hip.so (gfx950 only) in-tree with hipcc.

Replaces the reference's nvcc JIT/setup.py builds (gridencoder/backend.py:6-38, raymarching/setup.py:45-63):
one shared library with a C ABI (include/lzzx_nerf_hip.h), no torch headers, no CUDA path.

    python -m lzzx_nerf_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
SO = os.path.join(LIBDIR, "liblzzx_nerf_hip.so")
SOURCES = ["lz_grid.hip", "lz_encoders.hip", "lz_raymarch.hip", "lz_head.hip", "lz_head_bwd.hip", "lz_head_rec.hip", "lz_head_rec16.hip", "lz_head_gradw.hip", "lz_head_f16.hip", "lz_frame.hip", "lz_render.hip", "lz_linear.hip", "lz_torso.hip", "lz_audio.hip"]
# -ffp-contract=off: every FMA in the kernels is explicit, so results are bit-identical to the CPU checker
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    inc = os.path.join(ROOT, "include")
    d += [os.path.join(inc, f) for f in os.listdir(inc)]
    return d


def up_to_date():
    if not os.path.exists(SO):
        return False
    t = os.path.getmtime(SO)
    return all(os.path.getmtime(p) <= t for p in _deps())


def build(force=False, verbose=False):
    if not force and up_to_date():
        return SO
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
