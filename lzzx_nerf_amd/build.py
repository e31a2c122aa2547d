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
SOURCES = ["lz_grid.hip", "lz_encoders.hip", "lz_raymarch.hip", "lz_head.hip", "lz_head_bwd.hip", "lz_head_rec.hip", "lz_head_rec16.hip", "lz_head_gradw.hip", "lz_head_f16.hip", "lz_frame.hip", "lz_ngp.hip", "lz_render.hip", "lz_linear.hip", "lz_torso.hip", "lz_audio.hip"]
# -ffp-contract=off: every FMA in the kernels is explicit, so results are bit-identical to the CPU checker
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
         "-Rpass-analysis=kernel-resource-usage"]   # the remarks are parsed into lib/kernel_resources.json (registers, spills, LDS per kernel)
RESOURCES = os.path.join(LIBDIR, "kernel_resources.json")
# Per-file additions.  -fno-slp-vectorize on the training heads: at -O3 the SLP vectorizer pairs adjacent scalar f32 multiplies / adds of the
# backward chains into v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (47-71 of them per backward kernel), and packed f32 vector ops cost more
# issue time than the scalar pair they replace next to MFMA work (MI355X_MICROARCH.md; lz_head_gather.h).  Same-box A/B of the cfg3 step,
# three alternations: f32-exact 12.21 -> 12.08 ms, -O 5.88 -> 5.82; same bits (the lanes of a packed op are independent IEEE operations).
FILE_FLAGS = {"lz_head_rec.hip": ["-fno-slp-vectorize"], "lz_head_bwd.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def parse_resources(stderr):
    """clang's kernel-resource-usage remarks -> {mangled kernel: {vgprs, agprs, sgpr_spill, vgpr_spill, scratch, occupancy, lds}}"""
    import re
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill",
            "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
    for line in stderr.splitlines():
        m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*(Function Name|[A-Za-z ]+(?:\[[a-z/A-Z]+\])?): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None and k in keys:
            cur[keys[k]] = int(v)
    return out


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

    resources = {}

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        resources[src] = parse_resources(r.stderr)
        if verbose and r.stderr:
            print("\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l), file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    import json
    with open(RESOURCES, "w") as f:   # a register spill in a kernel tuned to a register limit is a 1.5x slowdown that no parity test sees:
        json.dump(resources, f, indent=1, sort_keys=True)   # tests/test_cabi.py checks this report
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return SO


def build_variant(name, extra_flags, only=None):
    """Experiment builds (tools/*): the sources in `only` (default: all) recompiled with `extra_flags` (-D switches), the rest taken from
    the regular build's objects, linked into lib/variants/<name>.so; run with LZZX_NERF_HIP_SO=<that path> (lzzx_nerf_amd/_lib.py)."""
    build()
    vdir = os.path.join(LIBDIR, "variants")
    odir = os.path.join(OBJDIR, "variants", name)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    hipcc = _hipcc()
    objs = []

    def one(src):
        if only is not None and src not in only:
            return os.path.join(OBJDIR, src.replace(".hip", ".o"))
        obj = os.path.join(odir, src.replace(".hip", ".o"))
        r = subprocess.run([hipcc] + FLAGS + FILE_FLAGS.get(src, []) + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        res = parse_resources(r.stderr)
        for k, v in res.items():
            if v.get("vgpr_spill", 0) or v.get("scratch", 0):
                print("variant %s: %s spills (vgpr_spill %d, scratch %d)" % (name, k, v.get("vgpr_spill", 0), v.get("scratch", 0)), file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, SOURCES))
    so = os.path.join(vdir, name + ".so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return so


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m lzzx_nerf_amd.build --variant NAME [--only a.hip,b.hip] -- -DFOO=1 ...
        i = sys.argv.index("--variant")
        only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
        extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
        print(build_variant(sys.argv[i + 1], extra, only))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
