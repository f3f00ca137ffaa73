"""Diagnostic (GPU): why is the drop-in step slower over 200 steps than over 20?  Per-10-step timings, allocator
segment counts, with and without the Python garbage collector."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "street-gaussians-ns_amd")]
import torch
from sgn_rast import scenes, step

dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)

def run(n, fused=False, tag=""):
    for _ in range(10):
        step.train_step(P, cam, w_img, w_a, fused=fused)
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    ts, t0 = [], time.perf_counter()
    for i in range(n):
        step.train_step(P, cam, w_img, w_a, fused=fused)
        if (i + 1) % 20 == 0:
            torch.cuda.synchronize()
            t1 = time.perf_counter(); ts.append(round(1e3 * (t1 - t0) / 20, 3)); t0 = t1
    s1 = torch.cuda.memory_stats()
    print(tag, "ms/step per 20:", ts, "| segments allocated during loop:",
          s1["segment.all.allocated"] - s0["segment.all.allocated"], "reserved MB:",
          s1["reserved_bytes.all.current"] >> 20, "gc counts", gc.get_count(), flush=True)

run(200, tag="dropin gc on ")
gc.disable()
run(200, tag="dropin gc off")
gc.enable(); gc.collect()
run(200, fused=True, tag="fused  gc on ")
gc.freeze()
run(200, tag="dropin frozen")
