"""GPU (-m gpu): the opt-in HIP-graph replay of the depth-rank launch chain (`SGN_HIP_GRAPHS=1`, csrc/api.cpp
sgn_graph_*): a child process with the switch on ranks the same buffers repeatedly — first call captures, later calls
replay — and changed contents / other buffers; every ranking must equal torch's stable sort.  (Measured in round 4:
no gain over the thirteen plain launches on this ROCm, `profiles/r04ab_hip_graphs_ab.log`; hence opt-in.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, ctypes as C
sys.path[:0] = [%r]
import torch
from sgn_rast import _lib as L
lib = L.load()
dev = "cuda"
g = torch.Generator().manual_seed(0)
n = 200_000
ok = True
def rank(depths, radii, gid, ws):
    L.check(lib.sgn_depth_rank(n, L.ptr(depths), L.ptr(radii), L.ptr(gid), L.ptr(ws), ws.numel(), L.sort_rank_mode(), L.stream_ptr()), "rank")
def expect(depths, radii):
    key = torch.where(radii > 0, depths, torch.full_like(depths, float("inf")))
    return torch.sort(key, stable=True).indices.to(torch.int32)
ws = L.workspace(lib.sgn_depth_rank_workspace_bytes(n), torch.device(dev))
depths = (torch.rand(n, generator=g) * 50 + 0.5).to(dev)
radii = (torch.rand(n, generator=g) > 0.2).to(torch.int32).to(dev)
gid = torch.empty(n, dtype=torch.int32, device=dev)
for it in range(4):                     # same buffers, new contents each time: capture once, then replays
    depths.copy_((torch.rand(n, generator=g) * 50 + 0.5).to(dev))
    rank(depths, radii, gid, ws)
    vis = int((radii > 0).sum())
    ok &= bool(torch.equal(gid[:vis], expect(depths, radii)[:vis]))
d2 = depths.clone(); gid2 = torch.empty_like(gid)       # other buffers: another graph
rank(d2, radii, gid2, ws)
ok &= bool(torch.equal(gid2[:vis], expect(d2, radii)[:vis]))
rank(depths, radii, gid, ws)                             # and back to the first one
ok &= bool(torch.equal(gid[:vis], expect(depths, radii)[:vis]))
torch.cuda.synchronize()
print("HIP_GRAPHS_OK" if ok else "HIP_GRAPHS_MISMATCH")
"""


def test_depth_rank_replayed_as_a_graph_equals_a_stable_sort():
    env = dict(os.environ, SGN_HIP_GRAPHS="1")
    src = CHILD % os.path.join(ROOT, "street-gaussians-ns_amd")
    out = subprocess.run([sys.executable, "-c", src], env=env, capture_output=True, text=True, timeout=600)
    assert "HIP_GRAPHS_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
