"""CPU-only: the test infrastructure itself under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5 - the
reference's own `cargo test` runs with debug assertions and overflow checks):
  * the oracle (oracle/plk_oracle.cpp + .inc files), driven by oracle/sanitize_main.cpp over every family of entry points the
    parity tests lean on;
  * the device arithmetic headers (plonky_amd/csrc/{fp,fp29,fz,ecz,glv}.cuh - exactly what the kernels compile), driven by
    tests/fp_host_sanitize_main.cpp on the host.
A finding aborts the driver (-fno-sanitize-recover); the test asserts a clean exit and the driver's own self-checks."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-pthread",
       "-Wall", "-Wno-unused-function"]


def _build_and_run(tmp_path, src, name, marker):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++"] + SAN + ["-o", exe, src])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert marker in out.stdout


def test_oracle_under_asan_ubsan(tmp_path):
    _build_and_run(tmp_path, os.path.join(ROOT, "oracle", "sanitize_main.cpp"), "oracle_sanitize", "sanitize_main: ok")


def test_device_arithmetic_headers_under_asan_ubsan(tmp_path):
    _build_and_run(tmp_path, os.path.join(ROOT, "tests", "fp_host_sanitize_main.cpp"), "fp_sanitize", "fp_host_sanitize: ok")
