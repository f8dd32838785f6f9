"""CPU: the host mirror's file layer (host/segment_file.h, host_index.cpp through libii2_host.so) under AddressSanitizer +
UBSan and under ThreadSanitizer - the tests of tests/test_segment_files_cpu.py (no device involved) run in a child interpreter
that loads an instrumented build of the library (make -C inverted_index_2_amd/csrc host_asan / host_tsan).  CPU box only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "inverted_index_2_amd", "csrc")


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.parametrize("kind,runtime,marker", [("asan", "libasan.so", "ERROR: AddressSanitizer"), ("tsan", "libtsan.so", "WARNING: ThreadSanitizer")])
def test_file_layer_under_sanitizer(kind, runtime, marker):
    rt = _runtime(runtime)
    if rt is None:
        pytest.skip(runtime + " not installed")
    if not os.path.exists(os.path.join(ROOT, "inverted_index_2_amd", "libii2_hip.so")):
        import __graft_entry__
        __graft_entry__.build()
    subprocess.check_call(["make", "-C", CSRC, "host_" + kind], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env["II2_HOST_LIB"] = os.path.join(ROOT, "inverted_index_2_amd", "libii2_host_%s.so" % kind)
    # (libstdc++ next to the runtime: the interpreter does not link it, and the runtime resolves __cxa_throw when it starts)
    cxx = _runtime("libstdc++.so.6") or _runtime("libstdc++.so")
    env["LD_PRELOAD"] = rt + (" " + cxx if cxx else "")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0"       # (the interpreter itself leaks by design)
    env["TSAN_OPTIONS"] = "report_signal_unsafe=0"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_segment_files_cpu.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert marker not in out and "runtime error:" not in out, out[-3000:]
