"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer runs (SURVEY.md 5 "race detection / sanitizers":
the reference has none).  (i) the oracle's C restatement -- every parity test leans on it --
(`make -C oracle sanitize`), (ii) the product's host-side graph analysis and strip schedule
(stereo_amd/csrc/trws_graph.cpp through tools/sanitize_graph.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_works(tmp_path):
    src = tmp_path / "t.c"
    src.write_text("int main(void){return 0;}\n")
    r = subprocess.run(["gcc", "-fsanitize=address,undefined", str(src), "-o", str(tmp_path / "t")], capture_output=True)
    return r.returncode == 0 and subprocess.run([str(tmp_path / "t")]).returncode == 0


def test_oracle_restatement_is_clean_under_sanitizers(tmp_path):
    if not shutil.which("gcc") or not _asan_works(tmp_path):
        pytest.skip("no working -fsanitize=address here")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SANITIZE_CHECK_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_host_graph_analysis_is_clean_under_sanitizers(tmp_path):
    if not shutil.which("g++") or not _asan_works(tmp_path) or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("no working -fsanitize=address / HIP headers here")
    exe = str(tmp_path / "sanitize_graph")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-w",
           os.path.join(ROOT, "tools", "sanitize_graph.cpp"), os.path.join(ROOT, "stereo_amd", "csrc", "trws_graph.cpp"),
           "-o", exe, "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SANITIZE_GRAPH_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
