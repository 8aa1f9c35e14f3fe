"""The C-ABI library loads and exports every symbol include/ptx.h declares; host-side error behaviour -- no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(ptx):
    assert declared_symbols() == sorted(ptx.binding.SYMBOLS)


def test_library_exports_every_declared_symbol(ptx):
    lib = ctypes.CDLL(ptx.lib_path())
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_struct_layouts_match_header(ptx):
    b = ptx.binding
    assert ctypes.sizeof(b.CParams) == 4 * 2 + 4 * 12 + 4 * 2
    assert ctypes.sizeof(b.CScene) == 16 + 8 * 11            # 3 x u32 (+4 pad) + 11 pointers
    assert ctypes.sizeof(b.CCounters) == 8 * 22


def test_no_cpu_fallback_without_gpu(ptx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ptx.PtxError) as e:
        ptx.Context(0)
    assert "device" in str(e.value).lower()


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "path-tracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".js", ".h", ".hip", ".c", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libptx_oracle" not in text and "ptx_oracle.h" not in text and "from oracle" not in text and "import oracle" not in text, f
