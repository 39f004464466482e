"""-m gpu: the C ABI from a plain C++ program (tests/c_abi/fused_l2_check.cpp): no Python objects, no
torch, its own hipMalloc'ed buffers and stream -- built with hipcc against include/mvin_hip.h and
libmvin_hip.so, checked against a direct double-precision evaluation of the equations."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fused_kernel_through_plain_cpp(hip_lib, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "fused_l2_check")
    lib_dir = os.path.join(ROOT, "mvin_amd")
    build = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", f"-I{os.path.join(ROOT, 'include')}",
                            os.path.join(ROOT, "tests", "c_abi", "fused_l2_check.cpp"), "-o", exe, f"-L{lib_dir}",
                            "-lmvin_hip", f"-Wl,-rpath,{lib_dir}"], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "C ABI OK" in run.stdout
