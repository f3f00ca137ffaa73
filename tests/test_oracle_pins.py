"""CPU: pins of the oracles against INDEPENDENT statements of the same mathematics (VERDICT r01 "weak #2": the
HIP kernel, the C oracle and the torch oracle all carry the same "fast" SH recurrence text, so their parity tests
compared a formula with itself; same single-source risk for the cube map).

* Spherical harmonics: the explicit real-SH polynomials with the constants of the 3D Gaussian Splatting reference
  implementation (C0, C1, C2[5], C3[7]; degree 4 from the orthonormal real SH definition via
  scipy's associated Legendre functions) in fp64, against both oracles' recurrence.
* Cube map: an fp64 brute-force sampler written from the OpenGL specification's cube-map face-selection table
  (sc, tc, ma per major axis) with its own seamless-edge handling (an (R+2)^2 apron per face filled from the
  neighbouring faces by 3-D re-projection of the apron texel's centre; corner apron texels averaged out), sharing no
  code with `oracle/torch_oracle.py:_cube_face_uv` / `_cube_dir` or the C restatement.

Still "parity unpinned" in the sense of the task (no upstream gsplat / nvdiffrast binary to run), but no longer
self-referential.
"""
import math

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O

# ---- 3DGS / gsplat constants, SURVEY.md A.6 (published values)
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def explicit_bases(d: np.ndarray) -> np.ndarray:
    """[n,3] (any length) -> [n,16] explicit polynomials b0..b15 of SURVEY.md A.6."""
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    b = [np.full_like(x, C0), -C1 * y, C1 * z, -C1 * x,
         C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy),
         C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
         C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
         C3[6] * x * (xx - 3 * yy)]
    return np.stack(b, axis=1)


def real_sh_from_definition(d: np.ndarray, degree: int) -> np.ndarray:
    """Orthonormal real spherical harmonics from their DEFINITION (associated Legendre functions, scipy), in the
    ordering and sign convention of the constants above: index l^2 + l + m, Y_lm ~ (-1)^m sqrt(2) K P_l^|m| cos/sin."""
    from scipy.special import lpmv
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    phi = np.arctan2(y, x)
    out = []
    for l in range(degree + 1):
        for m in range(-l, l + 1):
            am = abs(m)
            K = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - am) / math.factorial(l + am))
            P = lpmv(am, l, z)                      # includes the Condon-Shortley phase (-1)^m
            if m == 0:
                v = K * P
            elif m > 0:
                v = math.sqrt(2) * K * P * np.cos(am * phi)
            else:
                v = math.sqrt(2) * K * P * np.sin(am * phi)
            out.append(v)
    return np.stack(out, axis=1)


def _dirs(n=4000, seed=0):
    g = np.random.default_rng(seed)
    d = g.normal(size=(n, 3)) * g.uniform(0.1, 30.0, size=(n, 1))       # un-normalised on purpose
    return d


def test_sh_recurrence_equals_explicit_polynomials_fp64():
    d = _dirs()
    want = explicit_bases(d)
    got = torch.stack(O._sh_bases(torch.from_numpy(d), 3), dim=-1).numpy()
    assert got.shape == (d.shape[0], 16)
    assert np.abs(got - want).max() < 1e-14                             # measured 3e-15: the same polynomials


def test_sh_explicit_polynomials_are_the_real_sh_definition_up_to_degree_4():
    """The published constants ARE the orthonormal real SH in the convention that keeps the Condon-Shortley phase
    (b1 = -C1 y, b3 = -C1 x, ...) — and the oracles' degree-4 band, which SURVEY.md A.6 does not spell out as
    polynomials, matches the definition too."""
    d = _dirs(seed=1)
    defn = real_sh_from_definition(d, 4)
    ex = explicit_bases(d)
    assert np.abs(ex - defn[:, :16]).max() < 1e-13
    got = torch.stack(O._sh_bases(torch.from_numpy(d), 4), dim=-1).numpy()
    assert got.shape[1] == 25 and np.abs(got - defn).max() < 1e-12
    # orthonormality on the sphere (Monte-Carlo, 200k directions): the Gram matrix is the identity
    big = np.random.default_rng(2).normal(size=(200_000, 3))
    B = torch.stack(O._sh_bases(torch.from_numpy(big), 4), dim=-1).numpy()
    gram = 4 * math.pi * (B.T @ B) / big.shape[0]
    assert np.abs(gram - np.eye(25)).max() < 0.03


def test_c_oracle_and_fp32_torch_oracle_match_explicit_polynomials(c_oracle):
    d = _dirs(2000, seed=3).astype(np.float32)
    want = explicit_bases(d.astype(np.float64))
    coeffs = torch.zeros(d.shape[0], 16, 3)
    for k in range(16):                                 # colour = basis k: coefficient one-hot per call
        coeffs.zero_()
        coeffs[:, k, :] = 1.0
        c = c_oracle.sh_fwd(3, torch.from_numpy(d), coeffs)[:, 0].double().numpy()
        t = O.spherical_harmonics(3, torch.from_numpy(d), coeffs)[:, 0].double().numpy()
        assert np.abs(c - want[:, k]).max() < 2e-6, k   # fp32 evaluation of O(1) polynomials
        assert np.abs(t - want[:, k]).max() < 2e-6, k


# ------------------------------------------------------------------------------------------------ cube map
# OpenGL 4.6 core specification, table 8.19 ("Selection of cube map images"): major axis -> (target, sc, tc, ma)
def _gl_face_st(d):
    x, y, z = d
    ax, ay, az = abs(x), abs(y), abs(z)
    if ax >= ay and ax >= az:                           # ties go to x, then y (nvdiffrast's / the oracle's choice)
        face, sc, tc, ma = (0, -z, -y, ax) if x > 0 else (1, z, -y, ax)
    elif ay >= az:
        face, sc, tc, ma = (2, x, z, ay) if y > 0 else (3, x, -z, ay)
    else:
        face, sc, tc, ma = (4, x, -y, az) if z > 0 else (5, -x, -y, az)
    return face, 0.5 * (sc / ma + 1.0), 0.5 * (tc / ma + 1.0)


def _gl_dir(face, s, t):
    """Inverse of the table: the direction through (s, t) of `face` (ma = 1)."""
    sc, tc = 2.0 * s - 1.0, 2.0 * t - 1.0
    return [(1.0, -tc, -sc), (-1.0, -tc, sc), (sc, 1.0, tc), (sc, -1.0, -tc), (sc, -tc, 1.0), (-sc, -tc, -1.0)][face]


def brute_force_cube(tex: np.ndarray, dirs: np.ndarray) -> np.ndarray:
    """fp64, per-sample: bilinear lookup in a face with a one-texel apron taken from the adjacent faces (the texel
    whose centre the apron position re-projects to); the four apron corners have no texel — their weight is dropped
    and the rest renormalised (nvdiffrast's documented seamless behaviour)."""
    R, Cc = tex.shape[1], tex.shape[3]
    apron = np.full((6, R + 2, R + 2, Cc), np.nan)
    apron[:, 1:-1, 1:-1] = tex
    for f in range(6):
        for j in range(-1, R + 1):
            for i in range(-1, R + 1):
                inside_i, inside_j = 0 <= i < R, 0 <= j < R
                if inside_i and inside_j or (not inside_i and not inside_j):
                    continue                            # interior, or a corner (stays NaN = "no texel")
                f2, s2, t2 = _gl_face_st(_gl_dir(f, (i + 0.5) / R, (j + 0.5) / R))
                assert f2 != f
                apron[f, j + 1, i + 1] = tex[f2, min(R - 1, int(t2 * R)), min(R - 1, int(s2 * R))]
    out = np.zeros((dirs.shape[0], Cc))
    for n, d in enumerate(dirs):
        f, s, t = _gl_face_st(d)
        u, v = s * R - 0.5, t * R - 0.5
        i0, j0 = math.floor(u), math.floor(v)
        au, av = u - i0, v - j0
        acc, wsum = np.zeros(Cc), 0.0
        for dj, wj in ((0, 1 - av), (1, av)):
            for di, wi in ((0, 1 - au), (1, au)):
                tx = apron[f, j0 + dj + 1, i0 + di + 1]
                if np.isnan(tx[0]):
                    continue
                acc += wi * wj * tx
                wsum += wi * wj
        out[n] = acc / wsum if 0 < wsum < 1 else acc
    return out


@pytest.mark.parametrize("R", [1, 2, 5, 8])
def test_cube_map_oracles_equal_independent_fp64_brute_force(c_oracle, R):
    g = np.random.default_rng(R)
    tex = g.uniform(size=(6, R, R, 3))
    d = g.normal(size=(3000, 3))
    # stress the seams: directions near face edges and corners
    edge = g.normal(size=(1500, 3))
    k = g.integers(0, 3, size=1500)
    edge[np.arange(1500), k] = np.sign(edge[np.arange(1500), k]) * np.abs(edge).max(axis=1) * g.uniform(0.97, 1.0, 1500)
    corner = np.sign(g.normal(size=(500, 3))) * g.uniform(0.9, 1.0, size=(500, 3))
    dirs = np.concatenate([d, edge, corner])
    want = brute_force_cube(tex, dirs)
    t32 = torch.from_numpy(tex).float()
    got_t = O.cube_texture(t32, torch.from_numpy(dirs).float()).double().numpy()
    got_c = c_oracle.cube_texture(t32, torch.from_numpy(dirs).float()).double().numpy()
    # fp32 evaluation near a face edge can pick the other face for a direction within rounding of the edge: there
    # both answers agree to O(eps * R) anyway because the lookup is continuous across edges (that IS seamlessness)
    assert np.abs(got_t - want).max() < 5e-5 * max(1, R)
    assert np.abs(got_c - want).max() < 5e-5 * max(1, R)
