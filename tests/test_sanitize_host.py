"""AddressSanitizer + UBSan on the CPU side (the GPU pool offers no device sanitizer): the oracle and the pure-host parts
of libparmgmc_hip are compiled with gcc -fsanitize=address,undefined into one small program (tests/sanitize/host_san.c)
and run; any finding aborts it."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    exe = tmp_path / "host_san"
    csrc = ROOT / "parmgmc_amd" / "csrc"
    srcs = [ROOT / "tests" / "sanitize" / "host_san.c", ROOT / "oracle" / "pmg_oracle.c", csrc / "pmg_common.c", csrc / "pmg_diag.c", csrc / "pmg_parsor.c", csrc / "pmg_rowblock.c"]
    cmd = [gcc, "-std=gnu11", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", str(ROOT / "include"), *map(str, srcs), "-o", str(exe), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-ldl", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1", "PATH": "/usr/bin:/bin"}, timeout=120)
    assert run.returncode == 0 and "host_san ok" in run.stdout, (run.stdout[-2000:], run.stderr[-4000:])
