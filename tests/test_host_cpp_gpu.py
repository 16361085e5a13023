"""Builds and runs tests/cpp/host_parity.cpp: the C++ host mirror of the reference's classes
(easysfm_amd/host/esfm_host.hpp) over the C ABI, checked against the oracle on a real GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    import oracle
    oracle.build()
    exe = os.path.join(tmp, "host_parity")
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "host_parity.cpp"), "-o", exe,
           os.path.join(ROOT, "easysfm_amd", "libesfm_hip.so"), os.path.join(ROOT, "oracle", "libesfm_oracle.so"),
           "-Wl,-rpath," + os.path.join(ROOT, "easysfm_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-fopenmp"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_host_mirror_compiles(tmp_path):
    """CPU: the header-only host layer and its test compile and link against the C ABI."""
    _build(str(tmp_path))


@pytest.mark.gpu
def test_host_mirror_parity_on_gpu(tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "HOST PARITY OK" in r.stdout, r.stdout
