# r04o: SH over un-concatenated sub-models: tests, then the fused scene-graph step with / without it (same box)
mkdir -p gpurun_out/r04o
O=$PWD/gpurun_out/r04o
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_scene_graph_at_size.py tests/test_gpu_literal_golden.py -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | head
python - <<'PY'
import sys, time, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "street-gaussians-ns_amd")]
import torch
from sgn_rast import ops, scenes, step
dev = torch.device("cuda", 0)
cam, raw = scenes.make_scene("metric", device=dev)
models, poses, idft = scenes.make_scene_graph(raw["means"].shape[0], cam, n_objects=8, object_frac=0.1, device=dev)
Ms = [step.leaf_params(m) for m in models]
w_img, w_a = step.loss_weights(cam, seed=1000, device=dev)
def one(parts):
    for m in Ms:
        for p in m.values():
            p.grad = None
    out = step.render_scene_graph(Ms, poses, idft, cam, 3, 16, fused=True, sh_parts=parts)
    (((out.rgb * w_img).sum() + (out.alpha * w_a).sum() + (out.object_acc * w_a).sum()) / (cam.height * cam.width)).backward()
for rep in range(2):
    for parts in (False, True):
        for _ in range(15): one(parts)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): one(parts)
        torch.cuda.synchronize()
        print(f"fused scene graph, sh_parts={parts}: {(time.perf_counter() - t0) * 10:.3f} ms/step", flush=True)
PY
