"""tools/ holds ~100 one-shot probes and sweeps around the product (nothing tested them).  They need a GPU to RUN; this keeps
them from rotting silently: every Python script must compile and every shell script must parse."""
import glob
import os
import py_compile
import subprocess

from conftest import ROOT


def test_every_python_tool_compiles(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(files) > 40
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


def test_every_shell_tool_parses():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert len(files) > 10
    for f in files:
        p = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert p.returncode == 0, (f, p.stderr)
