"""Where does the HOST time of one drop-in step go?  cProfile over 30 steps (GPU waits show up under .item())."""
import cProfile, pstats, sys, io, time, torch
sys.path.insert(0, "street-gaussians-ns_amd")
from sgn_rast import scenes, step, _lib as L
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", seed=0, device=dev)
P = step.leaf_params(raw)
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
fused = len(sys.argv) > 1 and sys.argv[1] == "fused"
for _ in range(5):
    step.train_step(P, cam, w_img, w_a, 3, 16, fused=fused)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(30):
    step.train_step(P, cam, w_img, w_a, 3, 16, fused=fused)
torch.cuda.synchronize()
pr.disable()
print("ms/step under cProfile:", (time.perf_counter() - t0) / 30 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
