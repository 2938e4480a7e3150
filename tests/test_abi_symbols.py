"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/qmg_hip.h declares.  No compute call is made (there is no GPU here)."""
import ctypes
import importlib
import os
import re

import pytest

qmg = importlib.import_module("quantum-mg_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    qmg.build()
    return qmg.lib()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "qmg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qmg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(qmg.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(so):
    for name in header_symbols():
        assert hasattr(so, name), "libqmg_hip.so does not export %s" % name


def test_status_strings_and_version(so):
    assert so.qmg_status_string(0) == b"success"
    assert b"gfx950" in so.qmg_version()
    assert so.qmg_status_string(3) == b"unsupported"


def test_desc_struct_layout_matches_header():
    # int Lx,Ly,nc (+pad) ; 2 pointers ; 6 doubles
    assert ctypes.sizeof(qmg.StencilDesc) == 16 + 16 + 48
    assert qmg.StencilDesc.clover.offset == 16 and qmg.StencilDesc.shift.offset == 32


def test_library_has_gfx950_code_object():
    data = open(qmg.SO_PATH, "rb").read()
    assert b"gfx950" in data


def test_product_does_not_reference_the_oracle():
    """The product path must not include, link or import anything under oracle/."""
    pkg = os.path.join(ROOT, "quantum-mg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".hip", ".h", ".hpp", ".cpp", ".py", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "qmg_oracle" not in text and "oracle_lib" not in text, os.path.join(dirpath, f)
    out = os.popen("ldd %s" % qmg.SO_PATH).read()
    assert "oracle" not in out
