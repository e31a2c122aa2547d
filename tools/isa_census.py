#!/usr/bin/env python3
"""Instruction census of one kernel by source section: copies csrc/ to a scratch directory, plants `; LZMARK name` comments at the given
source anchors, compiles to ISA and counts instructions between the markers (the slices are straight-line code, so static = dynamic).
Markers move a little with instruction scheduling: read neighbouring sections together.

    python tools/isa_census.py            # the f16 fused frame kernel lz_k_frame<1, 1, 2>
"""
import collections
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lzzx_nerf_amd", "csrc")
TMP = "/tmp/lz_census"
_pos = [a for a in sys.argv[1:] if not a.startswith("-")]
KERNEL = _pos[0] if _pos else "_Z10lz_k_frameILi1ELi1ELi2EEvN7LzfHeadIXT_EE4ArgsE8LzFrameK"

MARKS = {
    "lz_head_f16w_slice.h": [("    lz_h8 bx[3];\n", "gather"), ("    lz_h8 att16[2];   // [u][j]", "aud"), ("    {   // ambient_aud = || att ||_2", "norm"),
                             ("    float eyeatt = 0.0f;", "eye"), ("    lz_h8 geo16[4];", "sigma"), ("        lz_h8 b1[6];\n", "colour"),
                             ("    out.eyeatt = eyeatt;", "end")],
    "lz_head_gather.h": [("    bool oobc[3];", "g_pos"), ("    uint32_t rowH[2][3][2]", "g_rows"), ("    float gv[9][4];", "g_loads"),
                         ('    asm volatile("" ::"v"(gv[0][0])', "g_pin"), ("    if constexpr (PACK) {", "g_interp")],
    "lz_frame.hip": [("            int ray = slot_lane ? sloti[SF_RAY * NS + sl] : -1;\n            bool have = false;", "F_refill_march"),
                     ("            typename HD::Out o;\n            if constexpr (PREC == 1) {", "F_head"), ("            __builtin_amdgcn_wave_barrier();     // the parked outputs", "F_composite")],
}


def main():
    shutil.rmtree(TMP, ignore_errors=True)
    shutil.copytree(SRC, TMP)
    for f, marks in MARKS.items():
        p = os.path.join(TMP, f)
        s = open(p).read()
        for anchor, name in marks:
            if anchor not in s:
                print("anchor not found:", f, name, file=sys.stderr)
                continue
            i = s.index(anchor)
            ls = s.rfind("\n", 0, i) + 1
            s = s[:ls] + '    asm volatile("; LZMARK %s");\n' % name + s[ls:]
        open(p, "w").write(s)
    out = os.path.join(TMP, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", os.path.join(TMP, "lz_frame.hip"), "-o", out], check=True, capture_output=True)
    s = open(out).read()
    i = s.index(KERNEL + ":")
    body = s[i:s.index(".Lfunc_end", i)].split("\n")
    sec, cnt, ops = "prologue", collections.OrderedDict(), collections.OrderedDict()
    for line in body:
        t = line.strip()
        m = re.match(r"; LZMARK (\w+)", t)
        if m:
            sec = m.group(1)
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        kind = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
                else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
        cnt.setdefault(sec, collections.Counter())[kind] += 1
        if kind == "valu":
            ops.setdefault(sec, collections.Counter())[op] += 1
    tot = collections.Counter()
    for k, v in cnt.items():
        print("%-16s %s" % (k, dict(v)))
        if k not in ("prologue", "F_refill_march", "F_composite"):
            tot.update(v)
    print("slice total (F_head .. end):", dict(tot))
    if "-v" in sys.argv:
        for k, v in ops.items():
            print(k, dict(v.most_common()))


if __name__ == "__main__":
    main()
