"""The diagnostic tooling of round 5 that runs without a GPU: the signal tracer preload
(tools/abort_trace.c) reports the native backtrace of the raising thread into the file named by
ABORT_TRACE_LOG even when file descriptor 2 points elsewhere (pytest's capture did exactly that to
the HSA runtime's fault message), and hands over to the handler that was installed before it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tracer(tmp_path):
    so = tmp_path / "libabort_trace.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-Wall", "-Werror", "-shared", "-fPIC",
                           os.path.join(ROOT, "tools", "abort_trace.c"), "-o", str(so)])
    return str(so)


def test_abort_trace_reports_into_its_own_file(tmp_path):
    so = _tracer(tmp_path)
    log = tmp_path / "trace.txt"
    code = ("import os, threading\n"
            "fd = os.open(os.devnull, os.O_WRONLY); os.dup2(fd, 2)\n"   # what a capture does to fd 2
            "t = threading.Thread(target=os.abort, name='worker'); t.start(); t.join()\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LD_PRELOAD=so, ABORT_TRACE_LOG=str(log),
                                                               ABORT_TRACE_MAPS=str(tmp_path / "maps.txt")))
    assert r.returncode == -6  # SIGABRT went on to the default action
    text = log.read_text()
    assert "abort_trace: signal 6" in text and "native backtrace" in text and "abort_trace: end" in text
    assert "libc" in text and "abort" in text          # the frames of the raising thread
    assert "pid" in text and "tid" in text
    maps = (tmp_path / "maps.txt").read_text()
    assert "libabort_trace.so" in maps and "[heap]" in maps


def test_abort_trace_chains_to_python_faulthandler(tmp_path):
    so = _tracer(tmp_path)
    log = tmp_path / "trace.txt"
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", "import os; os.abort()"],
                       env=dict(os.environ, LD_PRELOAD=so, ABORT_TRACE_LOG=str(log)), capture_output=True, text=True)
    assert r.returncode == -6
    assert "Fatal Python error: Aborted" in r.stderr       # faulthandler (installed later) ran first ...
    assert "abort_trace: signal 6" in log.read_text()      # ... and handed over to the preload
