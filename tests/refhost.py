"""Host harness that imports the REFERENCE's own model files, unchanged, from /root/reference (test infrastructure).

`load(backend)` puts `tests/stubs` (nerfstudio / kornia / pytorch3d / ... stand-ins) on `sys.path`, resolves the
three native-backed imports of `street_gaussians_ns/sgn_splatfacto.py:11-23` — `gsplat.*`, `pytorch_msssim`,
`nvdiffrast.torch` — to the chosen backend and imports `street_gaussians_ns.sgn_splatfacto` and
`street_gaussians_ns.sgn_splatfacto_scene_graph` from the read-only reference tree (bytecode writing disabled: nothing
is written under /root/reference, nothing is copied from it).

Backends:
* ``"oracle"`` — the CPU oracle behind the same operator surface (`tests/oracle_ops.py`, `oracle/torch_oracle.py`).
  This is what runs in this container (no GPU here): it proves that the reference's code runs, literally, on the
  surface the product implements, and produces the call trace the GPU replay is checked against.
* ``"hip"`` — the product's import shims (`street-gaussians-ns_amd/{gsplat,pytorch_msssim,nvdiffrast}`), for a
  machine that has both the reference checkout and an MI355X (the GPU box of this project has no /root/reference).

`cpu_as_cuda()` makes the reference's hard-coded ``device="cuda"`` / ``.cuda()`` spellings land on the CPU.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import types

import torch
from torch.overrides import TorchFunctionMode

HERE = os.path.dirname(os.path.abspath(__file__))
_SCRATCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_refscratch")     # tests/stage_reference.py
REFERENCE = os.environ.get("SGN_REFERENCE_ROOT") or ("/root/reference" if os.path.isdir("/root/reference") else _SCRATCH)
STUBS = os.path.join(HERE, "stubs")
_NATIVE = ("gsplat", "gsplat._torch_impl", "gsplat.project_gaussians", "gsplat.rasterize", "gsplat.sh", "gsplat.utils",
           "pytorch_msssim", "nvdiffrast", "nvdiffrast.torch")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE, "street_gaussians_ns", "sgn_splatfacto.py"))


class _CpuAsCuda(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device")
        if dev is not None and torch.device(dev).type == "cuda":
            kwargs["device"] = "cpu"
        if func is torch.Tensor.cuda:
            return args[0]
        if func is torch.Tensor.to and len(args) > 1 and isinstance(args[1], (str, torch.device)) \
                and torch.device(args[1]).type == "cuda":
            args = (args[0], "cpu") + tuple(args[2:])
        return func(*args, **kwargs)


@contextlib.contextmanager
def cpu_as_cuda():
    """Inside: tensor factories with device='cuda', `.cuda()`, `.to('cuda')` and `nn.Module.cuda()` stay on the CPU."""
    orig = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, device=None: self
    try:
        with _CpuAsCuda():
            yield
    finally:
        torch.nn.Module.cuda = orig


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def _oracle_backend() -> dict:
    import oracle_ops
    from oracle import torch_oracle as O

    class SSIM(torch.nn.Module):
        """pytorch_msssim.SSIM call shape on the oracle's restatement."""
        def __init__(self, data_range=255, size_average=True, win_size=11, win_sigma=1.5, channel=3, **kw):
            super().__init__()
            assert size_average and channel == 3
            self.data_range, self.win_size, self.win_sigma = data_range, win_size, win_sigma

        def forward(self, X, Y):
            return O.ssim(X, Y, self.data_range, self.win_size, self.win_sigma)

    def texture(tex, uv, filter_mode="linear", boundary_mode="cube"):
        assert filter_mode == "linear" and boundary_mode == "cube"
        return torch.stack([O.cube_texture(tex[b], uv[b]) for b in range(tex.shape[0])], 0)

    return {
        "gsplat": _module("gsplat"),
        "gsplat._torch_impl": _module("gsplat._torch_impl", quat_to_rotmat=O.quat_to_rotmat),
        "gsplat.project_gaussians": _module("gsplat.project_gaussians", project_gaussians=oracle_ops.project_gaussians),
        "gsplat.rasterize": _module("gsplat.rasterize", rasterize_gaussians=oracle_ops.rasterize_gaussians),
        "gsplat.sh": _module("gsplat.sh", num_sh_bases=O.num_sh_bases,
                             spherical_harmonics=oracle_ops.spherical_harmonics),
        "pytorch_msssim": _module("pytorch_msssim", SSIM=SSIM),
        "nvdiffrast": _module("nvdiffrast"),
        "nvdiffrast.torch": _module("nvdiffrast.torch", texture=texture),
    }


_loaded = {}


def load(backend: str = "oracle") -> types.SimpleNamespace:
    """Import the reference's model modules against ``backend``; cached per backend (one backend per process: the
    reference binds the operator names at import time)."""
    if backend in _loaded:
        return _loaded[backend]
    if _loaded:
        raise RuntimeError("the reference modules are already bound to another backend in this process")
    if not available():
        raise FileNotFoundError(REFERENCE)
    sys.dont_write_bytecode = True                      # never write __pycache__ under /root/reference
    saved = {k: sys.modules.get(k) for k in _NATIVE}
    if backend == "oracle":
        sys.modules.update(_oracle_backend())
    elif backend != "hip":
        raise ValueError(backend)
    for p in (STUBS, REFERENCE):
        if p not in sys.path:
            sys.path.append(p)                          # after the repo's own entries: shims keep priority
    try:
        with cpu_as_cuda() if backend == "oracle" else contextlib.nullcontext():
            splat = importlib.import_module("street_gaussians_ns.sgn_splatfacto")
            graph = importlib.import_module("street_gaussians_ns.sgn_splatfacto_scene_graph")
            annos = importlib.import_module("street_gaussians_ns.data.utils.dynamic_annotation")
    finally:
        for k, v in saved.items():                      # the product's shims stay what `import gsplat` resolves to
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    from nerfstudio.cameras.cameras import Cameras
    from nerfstudio.engine.callbacks import TrainingCallbackAttributes, TrainingCallbackLocation
    from nerfstudio.engine.optimizers import AdamOptimizerConfig, Optimizers
    ns = types.SimpleNamespace(splat=splat, graph=graph, annos=annos, Cameras=Cameras, Optimizers=Optimizers,
                               AdamOptimizerConfig=AdamOptimizerConfig,
                               TrainingCallbackAttributes=TrainingCallbackAttributes,
                               TrainingCallbackLocation=TrainingCallbackLocation, backend=backend)
    _loaded[backend] = ns
    return ns


def nerfstudio_camera(ns, cam, time=None):
    """`sgn_rast.scenes.Camera` (gsplat convention: +z forward, y down) -> nerfstudio `Cameras` (OpenGL c2w): the
    reference multiplies R by diag(1,-1,-1) at sgn_splatfacto.py:829-831, so the inverse flip goes in here."""
    w2c = cam.viewmat.detach().cpu().to(torch.float64)
    c2w_cv = torch.linalg.inv(w2c)
    c2w_gl = c2w_cv.clone()
    c2w_gl[:3, :3] = c2w_cv[:3, :3] @ torch.diag(torch.tensor([1.0, -1.0, -1.0], dtype=torch.float64))
    return ns.Cameras(c2w_gl[:3, :4].to(torch.float32), cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height,
                      times=time)


def set_gauss_params(model, raw: dict) -> None:
    """Overwrite a SplatfactoModel's parameters with a `scenes.make_gaussians` dictionary (names differ only)."""
    names = dict(means="means", scales="log_scales", quats="quats", features_dc="features_dc",
                 features_rest="features_rest", opacities="opacity_logits")
    for ref_name, ours in names.items():
        model.gauss_params[ref_name] = torch.nn.Parameter(raw[ours].detach().clone())


# ------------------------------------------------------------------------------------------------ model builders
def build_single(ns, raw: dict, sky_res: int = 8, step: int = 3000, **cfg_kw):
    """A literal `SplatfactoModel` (fourier dim 1: the stand-alone / background configuration, sgn_config.py:56)
    holding the Gaussians of ``raw``, in training mode at ``step`` (SH degree min(step // 1000, 3))."""
    with cpu_as_cuda() if ns.backend == "oracle" else contextlib.nullcontext():
        cfg = ns.splat.SplatfactoModelConfig(use_sky_sphere=sky_res > 0, env_map_res=max(sky_res, 1),
                                             fourier_features_dim=1, num_random=64, random_init=True, **cfg_kw)
        model = cfg.setup(scene_box=None, num_train_data=10)
        set_gauss_params(model, raw)
    model.train()
    model.step = step
    return model


def build_annotations(ns, models, poses, n_frames: int = 4, t0: float = 1.5e15):
    """`InterpolatedAnnotation` (the reference's own class) filled with one moving box per object model: box k of
    frame f sits at pose k's translation shifted by 0.1 (f - 2) m, orientation = pose k's rotation."""
    import numpy as np
    A = ns.annos
    annos = A.InterpolatedAnnotation(anno_json_path=None)
    stamps = [A.parse_timestamp(t0 + 1e5 * i) for i in range(n_frames)]
    for fi, ts in enumerate(stamps):
        boxes = []
        for k in range(1, len(models)):
            tid = f"t{k}"
            R = poses[k, :9].reshape(3, 3).double().cpu().numpy()
            t = poses[k, 9:12].double().cpu().numpy()
            box = A.Box(t + 0.1 * (fi - 2), trackId=tid, size=np.array([2.0, 1.5, 4.0]), label="car", frame_id=ts,
                        frame=fi, rot=R)
            boxes.append(box)
            if tid not in annos.objects_meta:
                annos.objects_meta[tid] = box
                annos.objects_frames[tid] = list(range(n_frames))
                n = models[k]["means"].shape[0]
                annos.seed_pts[tid] = (models[k]["means"].detach().cpu().clone(),
                                       torch.full((n, 3), 128.0))
        annos.annos[ts] = boxes
    annos.all_names = list(annos.annos.keys())
    annos.unique_track_ids = list(annos.objects_meta.keys())
    return annos, stamps


def build_scene_graph(ns, models, poses, sky_res: int = 8, step: int = 3000, fourier_dim: int = 5):
    """A literal `SplatfactoSceneGraphModel`: background = ``models[0]``, one object model per further entry, boxes from
    :func:`build_annotations`.  Returns (model, timestamps)."""
    annos, stamps = build_annotations(ns, models, poses)
    with cpu_as_cuda() if ns.backend == "oracle" else contextlib.nullcontext():
        bg = ns.splat.SplatfactoModelConfig(num_random=64, fourier_features_dim=1)
        obj = ns.splat.SplatfactoModelConfig(num_random=64, fourier_features_dim=fourier_dim)
        cfg = ns.graph.SplatfactoSceneGraphModelConfig(use_sky_sphere=sky_res > 0, env_map_res=max(sky_res, 1),
                                                       num_random=16, background_model=bg, object_model_template=obj,
                                                       fourier_features_dim=fourier_dim)
        model = cfg.setup(scene_box=None, num_train_data=10, metadata={"object_annos": annos})
        set_gauss_params(model.all_models["background"], models[0])
        for k in range(1, len(models)):
            set_gauss_params(model.all_models[f"object_t{k}"], models[k])
    model.train()
    model.step = step
    for m in model.all_models.values():
        m.train()
        m.step = step
    return model, stamps


def scene_graph_tables(ns, models, poses, frame: int, n_frames: int = 4, fourier_dim: int = 5):
    """(poses, idft) tables `sgn_rast.step.render_scene_graph` needs for the frame the literal model is shown:
    object k's translation as `build_annotations` placed it, and the reference's own IDFT weights."""
    p = poses.clone()
    p[1:, 9:12] = p[1:, 9:12] + 0.1 * (frame - 2)
    w = ns.graph.IDFT(float(frame / (n_frames - 1)), fourier_dim)[0]
    idft = torch.stack([torch.nn.functional.one_hot(torch.tensor(0), fourier_dim).float()] + [w] * (len(models) - 1))
    return p, idft
