import sys, torch, time
sys.path.insert(0, "street-gaussians-ns_amd")
from sgn_rast import sky, scenes, _lib as L
dev = "cuda"
cam = scenes.make_camera(1920, 1280, 2000.0, device=dev)
c2w = torch.zeros(3, 4, device=dev); c2w[:, :3] = cam.viewmat[:3, :3].T
base = (0.5 * torch.ones(6, 1024, 1024, 3, device=dev)).requires_grad_(True)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = sky.sky_color(base, 1280, 1920, 2000., 2000., 960., 640., c2w, None)
for name, g in [("zeros", torch.zeros_like(out)), ("ones", torch.ones_like(out)), ("rand", torch.rand_like(out))]:
    def f():
        base.grad = None
        out.backward(g, retain_graph=True)
    print("sky bwd total (memset+kernel)", name, round(timeit(f), 4), "ms")
print("sky fwd", round(timeit(lambda: sky.sky_color(base, 1280, 1920, 2000., 2000., 960., 640., c2w, None)), 4))
v = torch.empty_like(base)
print("memset only", round(timeit(lambda: v.zero_()), 4))
