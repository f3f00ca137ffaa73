"""Regenerates tests/golden/calltrace_{single,scene_graph}.json by running the REFERENCE's own model files
(imported from /root/reference through tests/refhost.py, CPU oracle backend) under the call tracer.
Run from the repo root in the build container:  python tests/golden/make_calltrace.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd"), os.path.join(ROOT, "tests")]

import refhost  # noqa: E402
import test_reference_literal as T  # noqa: E402
from calltrace import Tracer, canonical  # noqa: E402

ns = refhost.load("oracle")
single = T._trace_single_literal(ns)
tr = Tracer()
T._graph_literal(ns, tracer=tr)
for name, calls in (("single", single), ("scene_graph", tr.calls)):
    with open(os.path.join(HERE, f"calltrace_{name}.json"), "w") as f:
        f.write(canonical(calls) + "\n")
    print(name, len(calls), "calls")
