"""CPU tier, sanitizers (GPU AddressSanitizer is not available on this pool, so the host halves are what can be checked):
  * the CPU oracle rebuilt with -fsanitize=address,undefined (oracle/Makefile: libs2k_oracle_asan.so) runs the golden-vector
    suite -- the reference's own scalar code reads one byte past its input (src/nthash_hpc.rs:213-215); the
    restatement must not;
  * the C-ABI library rebuilt with host-side ASan (csrc/Makefile.asan: libs2k_asan.so: argument validation, FASTA/FASTQ parser,
    pinned-ring copy threads) runs the host-only tests (parser on well-formed and malformed files, exports, layouts).
Each runs in a child interpreter with the sanitizer runtime preloaded; a report makes the child exit non-zero."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, args, timeout=900):
    env = dict(os.environ, **env_extra)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1"  # the interpreter itself is not leak-clean
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert "passed" in p.stdout
    return p.stdout


def test_oracle_golden_suite_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libs2k_oracle_asan.so"])
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(rt):
        pytest.skip("gcc has no libasan here")
    out = _run({"LD_PRELOAD": rt, "S2K_ORACLE_LIB": os.path.join(ROOT, "oracle", "libs2k_oracle_asan.so")},
               ["tests/test_oracle_golden.py", "-m", "not gpu"])
    assert "failed" not in out


def test_host_side_of_the_library_under_asan():
    csrc = os.path.join(ROOT, "rust-seq2kminmers_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "-f", "Makefile.asan", "-j", "4", "libs2k_asan.so"])
    import glob

    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rts:
        pytest.skip("no clang ASan runtime in this ROCm")
    out = _run({"LD_PRELOAD": rts[0], "S2K_LIB": os.path.join(csrc, "libs2k_asan.so")},
               ["tests/test_fastx.py", "tests/test_abi.py", "-m", "not gpu"])
    assert "failed" not in out
