"""The host-side math of the hot path (per-pair Umeyama, the LUM loop with its banded AVX solve, the one-call host
step) under AddressSanitizer + UBSan.  GPU sanitizers are not available on the pool; the host code is what they can
cover, so it is compiled straight from csrc/host_math.cpp with the sanitizers on and driven over ring graphs of 2..36
views, a complete graph (fill-in) and an edge with too few correspondences."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs the ROCm host compiler")
def test_host_math_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "asan_host_math")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-march=x86-64-v3",
           "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-Wno-option-ignored",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "multi-view-registration_amd", "csrc"),
           "-x", "hip", os.path.join(ROOT, "tests", "cxx", "asan_host_math.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("V=") or l.startswith("complete")]
    assert len(lines) == 6 and all(" rc=0 " in l for l in lines), r.stdout


def test_shim_host_side_is_clean_under_asan_and_ubsan(tmp_path):
    """the C++ shim's host code -- PCD v0.7 reader / writer with the LZF codec, points.asc, transformation.txt /
    axis.txt, refineAxis -- is parser code fed with files: the same driver the format tests use, rebuilt with the
    sanitizers on (it links the product library for the host math)."""
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    pkg = os.path.join(ROOT, "multi-view-registration_amd")
    if not os.path.exists(os.path.join(pkg, "libmvr_hip.so")):
        pytest.skip("library not built")
    exe = str(tmp_path / "host_driver_asan")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "host_driver.cpp"), "-o", exe,
           "-L" + pkg, "-lmvr_hip", "-Wl,-rpath," + pkg]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    d = tmp_path / "files"
    d.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=120, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-2000:]


def test_pcd_reader_survives_damaged_files_under_asan(tmp_path):
    """4 500 mutated copies of small PCD files in the three encodings (flipped bits, edited header digits, truncations,
    insertions, splices) and headers that promise 10^11 points: the reader may accept or reject, it must not read or
    write outside its buffers nor size a buffer from a header alone (tests/cxx/pcd_fuzz.cpp)."""
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = str(tmp_path / "pcd_fuzz")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "pcd_fuzz.cpp"), "-o", exe],
                   check=True, capture_output=True, timeout=300)
    d = tmp_path / "files"
    d.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:max_allocation_size_mb=512",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-2000:]
    assert "accepted=" in r.stdout and "rejected=" in r.stdout
    assert "size/type combinations=6912" in r.stdout        # 2 encodings x 12^3 SIZE/TYPE triples x {z, z + rgb}
