"""CPU: the C-ABI library loads, exports every symbol include/ii2.h declares, and refuses
to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from inverted_index_2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "ii2.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ii2_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _header_symbols() == sorted(_lib.PROTOTYPES)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(_lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert _lib.load().ii2_abi_version() == 1


def test_library_exports_nothing_but_the_header():
    """libii2_hip.so's dynamic symbol table is exactly include/ii2.h (the host mirror lives in libii2_host.so)."""
    import subprocess
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip() and " T " in line or " W " in line})
    assert exported == _header_symbols(), sorted(set(exported) ^ set(_header_symbols()))


def test_host_mirror_is_a_separate_library_over_the_abi():
    from inverted_index_2_amd import host
    if not os.path.exists(host.HOST_LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(_lib.LIB_PATH)        # dependency first
    h = C.CDLL(host.HOST_LIB_PATH)
    assert hasattr(h, "ii2h_create") and not hasattr(lib, "ii2h_create")
    src = open(os.path.join(ROOT, "inverted_index_2_amd", "host", "host_index.cpp")).read()
    assert "internal.h" not in src and "hip/hip_runtime" not in src      # only include/ii2.h


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from inverted_index_2_amd import Context, II2Error
    with pytest.raises(II2Error) as e:
        Context(0)
    assert e.value.code == -7      # II2_ENODEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "inverted_index_2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f
