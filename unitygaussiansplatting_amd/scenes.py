"""Deterministic synthetic splat scenes + cameras for the BASELINE.json configs.

The INRIA "bicycle"/"garden" models are not available offline, so C2/C3 are *bicycle-sized* /
*garden-sized* synthetic stand-ins: same splat count, asset format, resolution and camera intrinsics
(fov 39.0965 deg, near 0.3, far 1000: /root/reference/projects/GaussianExample/Assets/GSTestScene.unity:277-279).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from .creator import InputSplatData

f32 = np.float32


@dataclass(frozen=True)
class SceneConfig:
    key: str
    label: str
    n: int
    seed: int
    extent: float           # splats live in [-extent, extent]^3
    surface_frac: float
    logscale_mu: float
    logscale_sigma: float
    quality: str            # creator quality preset
    width: int
    height: int
    fov_y: float
    eye_radius: float
    eye_elev_deg: float


CONFIGS: Dict[str, SceneConfig] = {
    "C1": SceneConfig("C1", "synthetic 100k splats, 640x360, Medium", 100_000, 1, 3.0, 0.60, -4.0, 0.8,
                      "Medium", 640, 360, 39.0965, 6.0, 0.0),
    "C2": SceneConfig("C2", "bicycle-sized synthetic 6,131,954 splats, 1200x797, Medium", 6_131_954, 2, 12.0, 0.75,
                      -4.6, 1.1, "Medium", 1200, 797, 39.0965, 8.0, 15.0),
    "C3": SceneConfig("C3", "garden-sized synthetic 5,834,784 splats, 1920x1080, VeryHigh fp32", 5_834_784, 3, 12.0,
                      0.75, -4.6, 1.1, "VeryHigh", 1920, 1080, 47.0, 8.0, 15.0),
    "C4": SceneConfig("C4", "synthetic 50M splats, 3840x2160, Medium", 50_000_000, 4, 40.0, 0.75, -4.6, 1.1,
                      "Medium", 3840, 2160, 60.0, 8.0, 15.0),
    # NOT a BASELINE.json configuration: C2's count / resolution / camera with larger splats (log-scale mean -3.0 instead of -4.6), so
    # that every visible splat lands on ~20 tiles like a trained outdoor scene does -- the regime in which the reference spends
    # 4.5 of its 6.8 ms in the rasteriser.  bench.py --config C2d; reported next to C2, never as the headline.
    "C2d": SceneConfig("C2d", "C2 with bicycle-like overdraw (log-scale mean -3.0): 6,131,954 splats, 1200x797, Medium [non-headline stress]",
                       6_131_954, 2, 12.0, 0.75, -3.0, 1.1, "Medium", 1200, 797, 39.0965, 8.0, 15.0),
    "C5": SceneConfig("C5", "bicycle-sized synthetic 6,131,954 splats, 8 cameras @1920x1080, Medium", 6_131_954, 2,
                      12.0, 0.75, -4.6, 1.1, "Medium", 1920, 1080, 39.0965, 8.0, 15.0),
}


def make_splats(n: int, seed: int, extent: float = 3.0, surface_frac: float = 0.6, logscale_mu: float = -4.0,
                logscale_sigma: float = 0.8) -> InputSplatData:
    """Raw (PLY-domain) splats: log-scales, opacity logits, (w,x,y,z) quaternions, f_dc / f_rest SH.

    Geometry: `surface_frac` of the splats lie on 6 noisy primitives (3 planes, 3 spheres, sigma = 0.02 of
    the extent-normalised unit), the rest is uniform in the box.  Appearance as SURVEY.md section 8d.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    ns = int(n * surface_frac)
    nv = n - ns
    E = f32(extent)
    pos = np.empty((n, 3), f32)
    prim = rng.integers(0, 6, size=ns)
    u = rng.random((ns, 3), dtype=f32) * 2 - 1                     # in [-1,1)
    noise = (rng.standard_normal((ns, 3), dtype=f32) * f32(0.02))
    p = np.empty((ns, 3), f32)
    # planes: z = -0.4 (ground, xy-plane in scene y-up => use y), x = 0.55, tilted plane
    m = prim == 0
    p[m] = np.stack([u[m, 0], np.full(m.sum(), -0.45, f32), u[m, 2]], 1)
    m = prim == 1
    p[m] = np.stack([np.full(m.sum(), 0.6, f32), u[m, 1] * f32(0.6), u[m, 2]], 1)
    m = prim == 2
    p[m] = np.stack([u[m, 0], u[m, 1] * f32(0.5), (u[m, 0] * f32(0.35) - f32(0.5)).astype(f32)], 1)
    # spheres
    centres = np.array([[0.0, 0.0, 0.0], [-0.45, 0.15, 0.3], [0.35, -0.2, -0.35]], f32)
    radii = np.array([0.33, 0.2, 0.14], f32)
    for k in range(3):
        m = prim == 3 + k
        d = rng.standard_normal((int(m.sum()), 3), dtype=f32)
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(f32)
        p[m] = centres[k] + d * radii[k]
    pos[:ns] = (p + noise) * E
    pos[ns:] = (rng.random((nv, 3), dtype=f32) * 2 - 1) * E
    np.clip(pos, -E, E, out=pos)

    logscale = rng.standard_normal((n, 3), dtype=f32) * f32(logscale_sigma) + f32(logscale_mu)
    np.clip(logscale, -8.0, -1.5, out=logscale)
    rot = rng.standard_normal((n, 4), dtype=f32)
    rot /= np.maximum(np.linalg.norm(rot, axis=1, keepdims=True), 1e-12).astype(f32)
    sel = rng.random(n) < 0.5
    opacity = np.where(sel, rng.standard_normal(n, dtype=f32) * f32(1.0) + f32(3.0),
                       rng.standard_normal(n, dtype=f32) * f32(1.5) - f32(2.0)).astype(f32)
    dc0 = rng.standard_normal((n, 3), dtype=f32) * f32(0.8) + f32(0.5)
    sh = rng.standard_normal((n, 15, 3), dtype=f32) * f32(0.08)
    # shuffle so that file order is not geometry order (the creator Morton-sorts anyway)
    perm = rng.permutation(n)
    return InputSplatData(pos[perm], dc0[perm], sh[perm], opacity[perm], logscale[perm], rot[perm])


def make_config_splats(cfg: SceneConfig, n_override: int = 0) -> InputSplatData:
    n = n_override or cfg.n
    return make_splats(n, cfg.seed, cfg.extent, cfg.surface_frac, cfg.logscale_mu, cfg.logscale_sigma)


def orbit_eye(radius: float, elev_deg: float, azim_deg: float) -> Tuple[float, float, float]:
    e, a = math.radians(elev_deg), math.radians(azim_deg)
    return (radius * math.cos(e) * math.sin(a), radius * math.sin(e), radius * math.cos(e) * math.cos(a))
