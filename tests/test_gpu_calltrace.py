"""GPU (-m gpu): the call-site replay running on the HIP ops hands the library exactly the call sequence the
reference's own model files produce (frozen in tests/golden/calltrace_*.json by tests/golden/make_calltrace.py from a
literal run of /root/reference's code; equality of literal run and replay is asserted on the CPU in
tests/test_reference_literal.py).  Same ops, argument shapes, dtypes, scalars and aliasing structure."""
import json
import os

import pytest

from calltrace import canonical

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _golden(name):
    with open(os.path.join(GOLDEN, f"calltrace_{name}.json")) as f:
        return canonical(json.load(f))


def test_replay_on_hip_ops_matches_the_reference_call_trace():
    import test_reference_literal as T
    from sgn_rast import ops, scenes
    ops.clear_binning_cache()
    assert canonical(T._trace_single_replay(ops, device="cuda")) == _golden("single")
    cam, models, poses = T._graph_scene()
    _, p0, idft0 = scenes.make_scene_graph(4000, cam, n_objects=3, object_frac=0.3, fourier_dim=5, seed=0,
                                           z_range=(1.0, 5.0))
    ops.clear_binning_cache()
    assert canonical(T._trace_graph_replay(None, ops, device="cuda", tables=(p0, idft0))) == _golden("scene_graph")
