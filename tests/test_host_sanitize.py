"""CPU: the library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer.

tests/host_sanitize/Makefile compiles every csrc/ translation unit --cuda-host-only with
-fsanitize=address,undefined and links it against a malloc-backed stand-in for the HIP runtime (hip_stub.cpp), so the
plan builders (weight packing, Winograd tile / table builders, column-block and tile-form rules, arena sizing, every
upload and clear), the launch arithmetic of the forwards and the C ABI's argument checks run on the CPU with every
"device" buffer bounds-checked.  tests/host_sanitize/sweep.py drives it over grids 16..128, joints 1..64, cameras
2..48 and the three model sizes (here in its quick form, ~40 s; the full sweep: `JH_SAN_QUICK=0`, ~10 min, run by hand
-- HISTORY.md records its result).  No GPU sanitizers exist on this pool; this is the host half only."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "host_sanitize")


def asan_runtime():
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


@pytest.mark.skipif(asan_runtime() is None or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"),
                    reason="needs the ROCm clang with its sanitizer runtimes")
def test_host_code_under_asan_ubsan():
    jobs = str(min(8, os.cpu_count() or 1))
    build = subprocess.run(["make", "-C", SAN, "-j" + jobs], capture_output=True, text=True, timeout=900)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-3000:]
    env = dict(os.environ, LD_PRELOAD=asan_runtime(), JH_SAN_QUICK=os.environ.get("JH_SAN_QUICK", "1"),
               ASAN_OPTIONS="detect_leaks=0:detect_odr_violation=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               JH_LIBRARY_PATH=os.path.join(SAN, "build", "libjarvis_hip_san.so"), OMP_NUM_THREADS="2")
    run = subprocess.run([sys.executable, os.path.join(SAN, "sweep.py")], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=1500)
    tail = run.stdout[-3000:] + run.stderr[-4000:]
    assert run.returncode == 0, tail
    assert "SANITIZER SWEEP OK" in run.stdout, tail
    for marker in ("AddressSanitizer", "runtime error:", "UndefinedBehaviorSanitizer"):
        assert marker not in run.stderr, tail
    assert "rejected 0" in run.stdout
