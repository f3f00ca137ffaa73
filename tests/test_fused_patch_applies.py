"""CPU: `integration/fused_callsites.patch` still applies to the reference checkout and is
what `integration/make_fused_patch.py` generates from it.  Skipped where /root/reference does not exist."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SGN_REFERENCE_SRC", "/root/reference")
REL = "street_gaussians_ns/sgn_splatfacto.py"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, REL)), reason="needs the reference checkout")


def test_patch_applies_cleanly_and_compiles(tmp_path):
    os.makedirs(tmp_path / "street_gaussians_ns")
    shutil.copyfile(os.path.join(REF, REL), tmp_path / REL)
    r = subprocess.run(["patch", "-p1", "-i", os.path.join(ROOT, "integration", "fused_callsites.patch")],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr
    src = open(tmp_path / REL).read()
    compile(src, REL, "exec")
    assert src.count("sgn_fused.") == 3 and "depths[:, None].repeat(1, 3)" not in src   # the depth pass is gone


def test_patch_is_what_the_generator_produces(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    import make_fused_patch as M
    import difflib
    src = open(os.path.join(REF, REL)).read()
    diff = "".join(difflib.unified_diff(src.splitlines(True), M.patched(src).splitlines(True), "a/" + REL, "b/" + REL, n=2))
    assert diff == open(os.path.join(ROOT, "integration", "fused_callsites.patch")).read()
