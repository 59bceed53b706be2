"""The C ABI from a host with no Python and no PyTorch: tests/c_host/conv_roundtrip.c (plain C: include/ursonet_hip.h + the HIP
runtime C API) is compiled with gcc on the GPU box and run against liburso_hip.so."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_runs_a_layer_through_the_library(tmp_path):
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / HIP runtime headers on this box")
    lib = os.path.join(ROOT, "ursonet_amd", "lib")
    assert os.path.exists(os.path.join(lib, "liburso_hip.so")), "liburso_hip.so is not built"
    exe = str(tmp_path / "conv_roundtrip")
    r = subprocess.run(["gcc", "-std=c99", "-O1", os.path.join(ROOT, "tests", "c_host", "conv_roundtrip.c"), "-I", os.path.join(ROOT, "include"),
                        "-I", "/opt/rocm/include", "-L", lib, "-L", "/opt/rocm/lib", "-lurso_hip", "-lamdhip64", "-lm",
                        "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C HOST OK" in r.stdout, r.stdout + r.stderr
