"""The C ABI from a program that has never heard of Python or torch (tests/cabi/consumer.cpp): built here with hipcc against
include/lzzx_nerf_hip.h and the in-tree library, run on the GPU.  The header itself must also parse as plain C."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "lzzx_nerf_amd", "lib")


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "use.c"
    src.write_text('#include "lzzx_nerf_hip.h"\nint main(void) { lz_head_params p; (void)p; return lz_abi_version() < 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)], check=True)


@pytest.mark.gpu
def test_cpp_consumer_runs(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "consumer"
    subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "consumer.cpp"),
                    "-L", LIBDIR, "-llzzx_nerf_hip", f"-Wl,-rpath,{LIBDIR}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cabi consumer ok" in out.stdout
