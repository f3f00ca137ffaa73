"""CPU: `integration/fused_callsites.patch` and `integration/fused_scene_graph.patch` still
apply to the reference checkout and are what `integration/make_fused_patch.py` generates from it.  Skipped where
/root/reference does not exist."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SGN_REFERENCE_SRC", "/root/reference")
REL = "street_gaussians_ns/sgn_splatfacto.py"
REL_SG = "street_gaussians_ns/sgn_splatfacto_scene_graph.py"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, REL)), reason="needs the reference checkout")


def test_patches_apply_cleanly_and_compile(tmp_path):
    os.makedirs(tmp_path / "street_gaussians_ns")
    for rel in (REL, REL_SG):
        shutil.copyfile(os.path.join(REF, rel), tmp_path / rel)
    for name in ("fused_callsites.patch", "fused_scene_graph.patch"):
        r = subprocess.run(["patch", "-p1", "-i", os.path.join(ROOT, "integration", name)],
                           cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr
    src = open(tmp_path / REL).read()
    compile(src, REL, "exec")
    assert src.count("sgn_fused.") == 3 and "depths[:, None].repeat(1, 3)" not in src   # the depth pass is gone
    sg = open(tmp_path / REL_SG).read()
    compile(sg, REL_SG, "exec")
    # the aggregation glue is gone from the step: no object2world_gs call, no Fourier sum, no per-sub-model concatenation
    body = sg[sg.index("    def get_outputs("):sg.index("    def get_loss_dict(")]
    assert "object2world_gs(" not in body and "aggregate_submodel_var(" not in sg[sg.index("def get_submodel_output"):sg.index("    def get_outputs(")]
    assert sg.count("sgn_fused.") == 4 and "id_range=id_range" in sg


def test_patches_are_what_the_generator_produces(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    import make_fused_patch as M
    import difflib
    for rel, fn, name in ((REL, M.patched, "fused_callsites.patch"), (REL_SG, M.patched_scene_graph, "fused_scene_graph.patch")):
        src = open(os.path.join(REF, rel)).read()
        diff = "".join(difflib.unified_diff(src.splitlines(True), fn(src).splitlines(True), "a/" + rel, "b/" + rel, n=2))
        assert diff == open(os.path.join(ROOT, "integration", name)).read(), name
