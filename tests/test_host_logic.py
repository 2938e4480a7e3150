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


def test_a_driver_that_dies_in_static_teardown_still_delivers_its_output(tmp_path):
    """Regression for the round-2 GPU run's SIGSEGV with EMPTY stdout and stderr: a fault in a linked library's static
    destructor -- after main() has returned -- used to take every buffered line of a piped stdout with it.  With drivers/driver_common.hpp
    (line-buffered stdout, fault handler) the same crash leaves the results in stdout and the signal, the phase it happened in and a
    backtrace in stderr; the exit status is still the signal's."""
    import subprocess
    lib = tmp_path / "libboom.so"
    exe = tmp_path / "fault_demo"
    pkg = os.path.join(ROOT, "quantum-mg_amd")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "exit_crash_demo", "lib.cpp"), "-o", str(lib)])
    subprocess.check_call(["g++", "-O2", "-g", "-rdynamic", "-std=c++11", os.path.join(ROOT, "tests", "host", "fault_demo.cpp"), "-o", str(exe), "-L" + str(tmp_path), "-lboom",
                           "-L" + pkg, "-lqmg_hip", "-Wl,-rpath," + str(tmp_path), "-Wl,-rpath," + pkg, "-pthread"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == -11
    assert "result line 1\nresult line 2\n" in out.stdout and "[QMG-PHASE]: solve" in out.stdout
    assert "[QMG-FATAL]: signal 11 (SIGSEGV) in phase 'exit" in out.stderr and "backtrace" in out.stderr
    # the same program without the guard: nothing arrives (what the round-2 record looked like)
    plain = tmp_path / "plain"
    subprocess.check_call(["g++", "-O2", os.path.join(ROOT, "tools", "exit_crash_demo", "main.cpp"), "-o", str(plain), "-L" + str(tmp_path), "-lboom", "-Wl,-rpath," + str(tmp_path)])
    out = subprocess.run([str(plain)], capture_output=True, text=True, timeout=60)
    assert out.returncode == -11 and out.stdout == "" and out.stderr == ""
