"""CPU: the library's host-only entry points survive junk arguments (tests/host_fuzz.py), each family in a process of its own."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("family,scale", [("dci", 0.1), ("tables", 0.2), ("dmrs", 0.2), ("dl", 0.3), ("ul", 0.3), ("prach", 0.3), ("ctrl", 0.3), ("turbo", 0.3)])
def test_host_entry_points_survive_junk_arguments(family, scale):
    r = subprocess.run([sys.executable, os.path.join(HERE, "host_fuzz.py"), "7", family, str(scale)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "survived" in r.stdout, (family, r.returncode, r.stdout[-300:], r.stderr[-600:])
