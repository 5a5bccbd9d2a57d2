"""The C-ABI library loads and exports every symbol include/dvn_prover.h declares; host-only entry points
behave; device entry points fail loudly (no fallback) when no GPU is visible."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dvn_prover.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.load_library()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/dvn_prover.h but not exported"


def test_binding_covers_header(pkg):
    from davinci_node_amd import binding
    assert sorted(binding.EXPORTED_SYMBOLS) == header_symbols()


def test_version_and_plan_are_host_only(pkg):
    L = pkg.load_library()
    assert b"gfx950" in L.dvn_version()
    c, w = C.c_uint32(), C.c_uint32()
    assert L.dvn_msm_plan(1 << 22, C.byref(c), C.byref(w)) == 0
    assert c.value == 16 and w.value == 16           # 16 windows of 16 bits cover the 254-bit scalar + carry
    assert L.dvn_msm_plan(1 << 10, C.byref(c), C.byref(w)) == 0
    assert w.value == 254 // c.value + 1


def test_no_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu suite")
    with pytest.raises(pkg.DvnError) as e:
        pkg.Context(0)
    assert e.value.code == -3        # DVN_ERR_NO_DEVICE; the caller (Go shim) falls back to its own CPU prover


def test_null_arguments_rejected(pkg):
    L = pkg.load_library()
    assert L.dvn_init(None, 0, None) == -1
    assert L.dvn_msm_plan(16, None, None) == -1
    assert L.dvn_malloc(None, 16, None) == -1


def test_header_is_plain_c():
    """The drop-in boundary is a C ABI (cgo consumes the header): it must compile as C99 with no C++ or torch types."""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "hdr.c")
        open(src, "w").write('#include "dvn_prover.h"\nint main(void) { return 0; }\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-fsyntax-only", src])
