"""On-disk formats either side of the hot path: the INRIA-layout PLY the reference exports and
the parameter naming of its checkpoints.  Plain numpy / torch host code (no device work).

* :func:`write_ply` / :func:`read_ply` — `scripts/exporter.py:59-128` (`ExportGaussianSplat.save_gs_model`): one
  `vertex` element of float32 properties in the order ``x y z nx ny nz f_dc_0..2 f_rest_0..3(K-1)-1 opacity scale_0..2
  rot_0..3``; normals zero; ``f_dc`` = Fourier coefficient 0 of ``features_dc`` (``shs_0``, `sgn_splatfacto.py:340-343`);
  ``f_rest`` channel-major (``shs_rest.transpose(1, 2)``, the INRIA order); opacity and scales RAW (logit / log), as the
  reference writes them; rows with a non-finite value are dropped.  Written as ``binary_little_endian`` without the
  ``plyfile`` dependency (the byte layout is the same).
* :func:`model_state` / :func:`load_model_state` — the key names of the reference's ``state_dict``
  (``gauss_params.means`` ... ``gauss_params.opacities``; sub-models of the scene graph under
  ``all_models.<name>.``) and its size-changing load (`sgn_splatfacto.py:425-437`: parameters are re-allocated to the
  checkpoint's number of Gaussians before loading; `sgn_splatfacto_scene_graph.py:393-400`: keys are routed to the
  sub-models by name).  `tests/test_reference_literal.py` loads a state written here into the reference's own model.
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np
import torch

# our parameter names (sgn_rast.step) -> the reference's gauss_params names
REF_NAMES = dict(means="means", log_scales="scales", quats="quats", features_dc="features_dc",
                 features_rest="features_rest", opacity_logits="opacities")


def ply_fields(n_rest_coeffs: int):
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    names += [f"f_rest_{i}" for i in range(3 * n_rest_coeffs)]
    return names + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]


def write_ply(path: str, params: Mapping[str, torch.Tensor]) -> int:
    """Returns the number of Gaussians written (non-finite rows are dropped, as the reference does)."""
    g = {k: v.detach().cpu().float().numpy() for k, v in params.items()}
    n = g["means"].shape[0]
    cols = [g["means"], np.zeros_like(g["means"]), g["features_dc"][:, 0, :],
            np.ascontiguousarray(g["features_rest"].transpose(0, 2, 1)).reshape(n, -1),
            g["opacity_logits"].reshape(n, 1), g["log_scales"], g["quats"]]
    table = np.concatenate(cols, axis=1).astype("<f4")
    table = table[np.isfinite(table).all(axis=1)]
    fields = ply_fields(g["features_rest"].shape[1])
    assert table.shape[1] == len(fields)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {table.shape[0]}"]
    header += [f"property float {f}" for f in fields] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(table.tobytes())
    return int(table.shape[0])


def read_ply(path: str) -> Dict[str, torch.Tensor]:
    """Inverse of :func:`write_ply` (binary little-endian float32 vertex element in the field order above)."""
    with open(path, "rb") as f:
        fields, n = [], 0
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property"):
                typ, name = line.split()[1:3]
                if typ not in ("float", "float32"):
                    raise ValueError(f"unsupported property type {typ}")
                fields.append(name)
            elif line.startswith("format") and "binary_little_endian" not in line:
                raise ValueError("only binary_little_endian PLY files are read")
            elif line == "end_header":
                break
        table = np.frombuffer(f.read(4 * n * len(fields)), dtype="<f4").reshape(n, len(fields))
    col = {name: i for i, name in enumerate(fields)}
    n_rest = sum(1 for name in fields if name.startswith("f_rest_")) // 3
    if fields != ply_fields(n_rest):
        raise ValueError("not an INRIA-layout Gaussian-splat PLY")
    take = lambda names: torch.from_numpy(np.stack([table[:, col[k]] for k in names], axis=1).copy())
    rest = take([f"f_rest_{i}" for i in range(3 * n_rest)]).reshape(n, 3, n_rest).transpose(1, 2).contiguous()
    return dict(means=take(["x", "y", "z"]), features_dc=take(["f_dc_0", "f_dc_1", "f_dc_2"])[:, None, :],
                features_rest=rest, opacity_logits=take(["opacity"]), log_scales=take(["scale_0", "scale_1", "scale_2"]),
                quats=take(["rot_0", "rot_1", "rot_2", "rot_3"]))


def model_state(models: Mapping[str, Mapping[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """``{"": params}`` for a single model, ``{"background": ..., "object_<id>": ...}`` for a scene graph ->
    flat state dictionary with the reference's key names."""
    out = {}
    for name, params in models.items():
        prefix = f"all_models.{name}." if name else ""
        for ours, ref in REF_NAMES.items():
            out[f"{prefix}gauss_params.{ref}"] = params[ours].detach().cpu().clone()
    return out


def load_model_state(state: Mapping[str, torch.Tensor], device="cpu") -> Dict[str, Dict[str, torch.Tensor]]:
    """Inverse of :func:`model_state`; every model comes back with the checkpoint's number of Gaussians, whatever the
    size of the model it is loaded into was (the reference re-allocates before loading, `sgn_splatfacto.py:425-437`).
    Also accepts the old flat names (``means`` ...) the reference remaps at `:428-432`."""
    models: Dict[str, Dict[str, torch.Tensor]] = {}
    back = {v: k for k, v in REF_NAMES.items()}
    for key, t in state.items():
        parts = key.split(".")
        if parts[0] == "all_models":
            name, rest = parts[1], parts[2:]
        else:
            name, rest = "", parts
        if rest[0] == "gauss_params":
            rest = rest[1:]
        if len(rest) != 1 or rest[0] not in back:
            continue                                   # not a Gaussian parameter (sky sphere, buffers ...)
        models.setdefault(name, {})[back[rest[0]]] = t.detach().to(device).clone().requires_grad_(True)
    for name, p in models.items():
        if set(p) != set(REF_NAMES):
            raise KeyError(f"model '{name}': incomplete parameter set {sorted(p)}")
        n = p["means"].shape[0]
        if any(v.shape[0] != n for v in p.values()):
            raise ValueError(f"model '{name}': parameters disagree on the number of Gaussians")
    return models
