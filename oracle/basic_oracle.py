"""CPU definition of the flatfield ESTIMATE -- test infrastructure, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` may import anything under
``oracle/``; the product path (image-stitcher_amd/) never does.

PARITY UNPINNED.  The reference estimates its gains with a third-party package,
``basicpy.BaSiC(get_darkfield=False, smoothness_flatfield=1).fit(images).flatfield``
(/root/reference/stitcher.py:374-377; basicpy is un-pinned in install_requirements.sh and absent offline, as is
its jax dependency), on at most 32 randomly chosen tiles per timepoint, stopping once more than 48 are collected
(stitcher.py:381-395).  Neither basicpy nor any fixture of its output exists in this container, so nothing here can
be checked against it.  What follows restates the PUBLISHED algorithm -- BaSiC, Peng et al., Nat. Commun. 8:14836
(2017): the image stack is modelled as  I_i(x) = b_i * S(x) + R_i(x)  with a flatfield S that is sparse in the DCT
domain and a residual R that is sparse in the image domain, solved by a linearised alternating-direction method with
adaptive penalty (LADMAP, the fitting mode basicpy uses by default) inside an iteratively re-weighted L1 loop -- in
the configuration the reference asks for: no darkfield, flatfield smoothness weight 1, basicpy's documented
defaults for everything else (working size 128, epsilon 0.1, rho 1.5, mu_coef 12.5, max_mu_coef 1e7,
optimization_tol 1e-3, reweighting_tol 1e-2, max_iterations 500, max_reweight_iterations 10).  The device version
(csrc/basic.hip) is tested against THIS definition and against a planted gain, not against basicpy.
"""
from __future__ import annotations

import numpy as np

WORKING_SIZE = 128
EPSILON = 0.1
RHO = 1.5
MU_COEF = 12.5
MAX_MU_COEF = 1e7
OPTIMIZATION_TOL = 1e-3
REWEIGHTING_TOL = 1e-2
MAX_ITERATIONS = 500
MAX_REWEIGHT_ITERATIONS = 10


def resize_matrix(n_out: int, n_in: int) -> np.ndarray:
    """[n_out, n_in] weights of a separable linear (triangle-kernel) resampling with half-pixel centres; when
    shrinking, the kernel is widened by the scale factor (anti-aliasing), rows are normalised to sum 1 (edges
    included).  This is what ``jax.image.resize(method='linear')`` -- basicpy's resize -- computes per axis."""
    scale = n_in / n_out
    width = max(scale, 1.0)
    out_centres = (np.arange(n_out, dtype=np.float64) + 0.5) * scale
    in_centres = np.arange(n_in, dtype=np.float64) + 0.5
    d = np.abs(out_centres[:, None] - in_centres[None, :]) / width
    w = np.maximum(0.0, 1.0 - d)
    w /= w.sum(axis=1, keepdims=True)
    return w.astype(np.float32)


def resize(images: np.ndarray, h_out: int, w_out: int) -> np.ndarray:
    """[..., h, w] -> [..., h_out, w_out] float32 with ``resize_matrix`` along both axes (rows first)."""
    a = np.asarray(images, dtype=np.float32)
    ry = resize_matrix(h_out, a.shape[-2])
    rx = resize_matrix(w_out, a.shape[-1])
    return np.einsum('oy,...yx,px->...op', ry, a, rx, optimize=True).astype(np.float32)


def dct_matrix(n: int) -> np.ndarray:
    """Orthonormal DCT-II matrix C: dct(x) = C x, idct(y) = C^T y."""
    k = np.arange(n, dtype=np.float64)[:, None]
    j = np.arange(n, dtype=np.float64)[None, :]
    c = np.sqrt(2.0 / n) * np.cos(np.pi * (2 * j + 1) * k / (2 * n))
    c[0] = np.sqrt(1.0 / n)
    return c.astype(np.float32)


def shrink(x: np.ndarray, t) -> np.ndarray:
    """Soft threshold: sign(x) * max(|x| - t, 0)."""
    return np.sign(x) * np.maximum(np.abs(x) - t, 0.0)


def ladmap_fit(im: np.ndarray, w: np.ndarray, smoothness_flatfield: float = 1.0):
    """One LADMAP solve of   min  smoothness * |dct2(S)|_1 + |W o R|_1   s.t.  I_i = b_i S + R_i   (no darkfield).
    ``im``, ``w``: [n, h, w] float32.  Returns (S, R, b, iterations)."""
    n, hh, ww = im.shape
    cy, cx = dct_matrix(hh), dct_matrix(ww)
    f32 = np.float32
    s = np.median(im, axis=0).astype(f32)
    b = np.ones(n, dtype=f32)
    r = np.zeros_like(im)
    y = np.zeros_like(im)
    spectral_norm = np.linalg.norm(im.reshape(n, -1).astype(np.float64), ord=2)
    image_norm = f32(np.linalg.norm(im.astype(np.float64)))
    mu = f32(MU_COEF / spectral_norm)
    max_mu = f32(mu * MAX_MU_COEF)
    it = 0
    for it in range(1, MAX_ITERATIONS + 1):
        i_b = b[:, None, None] * s[None]
        eta = f32(np.sum(b * b) * 1.02 + 0.01)
        s_lin = s + np.sum(b[:, None, None] * (im - i_b - r + y / mu), axis=0) / eta
        s_new = cy.T @ shrink(cy @ s_lin @ cx.T, f32(smoothness_flatfield) / (eta * mu)) @ cx
        ds = s_new - s
        s = s_new.astype(f32)
        i_b = b[:, None, None] * s[None]
        r_new = shrink(im - i_b + y / mu, w / mu)
        dr = r_new - r
        r = r_new.astype(f32)
        resid = im - r
        b_new = np.maximum(np.sum(s[None] * (resid + y / mu), axis=(1, 2)) / np.sum(s * s), 0.0).astype(f32)
        db = b_new - b
        b = b_new
        fit = resid - b[:, None, None] * s[None]
        y = (y + mu * fit).astype(f32)
        change = max(float(np.sqrt(eta)) * float(np.linalg.norm(ds)), float(np.linalg.norm(dr)),
                     float(np.linalg.norm(s)) * float(np.linalg.norm(db))) / float(image_norm)
        residual = float(np.linalg.norm(fit.astype(np.float64))) / float(image_norm)
        if mu * change < 1e-2 * OPTIMIZATION_TOL * 10:      # the iterates have settled at this penalty: raise it
            mu = f32(min(mu * RHO, max_mu))
        if residual <= OPTIMIZATION_TOL and change <= OPTIMIZATION_TOL:
            break
    return s, r, b, it


def basic_fit(images: np.ndarray, smoothness_flatfield: float = 1.0, working_size: int = WORKING_SIZE):
    """BaSiC(get_darkfield=False, smoothness_flatfield=...).fit(images).flatfield -> [H, W] float32, mean ~1.
    ``images``: [n, H, W] of any real dtype."""
    images = np.asarray(images)
    if images.ndim != 3:
        raise ValueError(f"images must be (N, Y, X), got {images.shape}")
    n, hh, ww = images.shape
    im = resize(images.astype(np.float32), working_size, working_size)
    w = np.ones_like(im)
    last = None
    s = b = None
    info = []
    for _ in range(MAX_REWEIGHT_ITERATIONS):
        s, r, b, its = ladmap_fit(im, w, smoothness_flatfield)
        mean_s = np.float32(s.mean())
        s = s / mean_s
        b = b * mean_s
        i_b = b[:, None, None] * s[None]
        w = 1.0 / (np.abs(r / (i_b + np.float32(EPSILON))) + np.float32(EPSILON))
        w = (w / w.mean()).astype(np.float32)
        info.append(its)
        if last is not None and float(np.mean(np.abs(s - last))) / float(np.mean(np.abs(last))) <= REWEIGHTING_TOL:
            break
        last = s
    return resize(s, hh, ww), {'ladmap_iterations': info, 'baseline': b}
