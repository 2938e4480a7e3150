"""CPU-side checks of the C++ facade's solver bookkeeping (tests/host/host_logic.cpp): the raw-direction GCR
back-substitution, the zero-guess hint, batch masks / views.  Compiled with g++ against the headers; libqmg_hip.so is
linked for its symbols only -- no GPU call is made."""
import importlib
import os
import subprocess

qmg = importlib.import_module("quantum-mg_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_facade_host_logic(tmp_path):
    qmg.build()
    libdir = os.path.join(ROOT, "quantum-mg_amd")
    exe = str(tmp_path / "host_logic")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-Wall", "-Wno-unused-variable", "-Wno-unused-parameter", "-o", exe,
                           os.path.join(ROOT, "tests", "host", "host_logic.cpp"), "-L" + libdir, "-lqmg_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host logic ok" in out.stdout
