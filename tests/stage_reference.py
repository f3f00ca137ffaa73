"""Stage the reference's model files for ONE `gpurun` call (test infrastructure; VERDICT r02 "missing #2").

    python tests/stage_reference.py stage    # copy the files below into tests/_refscratch/ (git-ignored, untracked)
    python tests/stage_reference.py clean    # remove the scratch copy again

The GPU box has no /root/reference; git-ignored files travel with the `gpurun` snapshot, so the eight files the
reference's two model modules import ride in an untracked scratch directory and `SGN_REFERENCE_ROOT=tests/_refscratch`
points `tests/refhost.py` at them (`tests/test_gpu_reference_literal.py`).  The copy is removed right after the call:
nothing of the reference is committed or left in the working tree.
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("SGN_REFERENCE_SRC", "/root/reference")
DST = os.path.join(HERE, "_refscratch")
FILES = [
    "street_gaussians_ns/__init__.py",
    "street_gaussians_ns/sgn_splatfacto.py",
    "street_gaussians_ns/sgn_splatfacto_scene_graph.py",
    "street_gaussians_ns/data/__init__.py",
    "street_gaussians_ns/data/utils/__init__.py",
    "street_gaussians_ns/data/utils/bbox_optimizers.py",
    "street_gaussians_ns/data/utils/data_utils.py",
    "street_gaussians_ns/data/utils/dynamic_annotation.py",
]


def main(cmd: str) -> None:
    if cmd == "clean":
        shutil.rmtree(DST, ignore_errors=True)
        return
    assert cmd == "stage", cmd
    for rel in FILES:
        dst = os.path.join(DST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(SRC, rel), dst)
    print(f"staged {len(FILES)} files under {DST}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "stage")
