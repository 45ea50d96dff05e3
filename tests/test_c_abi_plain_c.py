"""The C ABI from PLAIN C: include/litho_abbe.h must be valid C99 (a binding in any language starts from it), and
examples/c_abi_demo.c -- the reference's demo configuration driven through liblitho_abbe.so without Python or torch -- must
compile and link against the library (CPU) and reproduce the reference's demo image on the GPU (`-m gpu`)."""
import os
import re
import shutil
import subprocess

import pytest

from helpers import ROOT

INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "lithographysimulator_amd", "lib")
DEMO = os.path.join(ROOT, "examples", "c_abi_demo.c")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "liblitho_abbe.so")):
        subprocess.check_call(["make", "-C", ROOT, "-j", "8", "all"])
    exe = str(tmp_path / "c_abi_demo")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", f"-I{ROCM}/include", f"-I{INC}", DEMO,
           f"-L{LIBDIR}", "-llitho_abbe", f"-L{ROCM}/lib", "-lamdhip64", "-lm", "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return exe


def test_header_is_valid_c99():
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    out = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", os.path.join(INC, "litho_abbe.h")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_plain_c_demo_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_plain_c_demo_reproduces_the_reference_demo_image(tmp_path):
    """64 x 64 demo: S = 184 source points, image sum 2.2029254e13 (the reference's own run, SURVEY.md 3.1 / golden g5); and a
    1000 x 1000 mask -- not a power of two -- runs embedded at 1024 on the coarse grid, from C, with nothing but the
    workspace size telling the caller so."""
    exe = _build(tmp_path)
    env = dict(os.environ, LD_LIBRARY_PATH=f"{LIBDIR}:{ROCM}/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("pn ")][0]
    assert " S 184 " in line and "image 64 x 64" in line and "runs at 64" in line, line
    total = float(re.search(r"sum ([-+.\de]+)", line).group(1))
    assert abs(total / 2.2029254e13 - 1) < 1e-5, line
    out = subprocess.run([exe, "1000"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("pn ")][0]
    assert "N 2048" in line and "runs at 1024" in line and "coarse grid 1" in line and "image 1000 x 1000" in line, line
    assert "k_ypass_rect<10, 8, true, 2>" in line and float(re.search(r"sum ([-+.\de]+)", line).group(1)) > 0, line
