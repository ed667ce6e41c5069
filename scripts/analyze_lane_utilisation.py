"""How many of the blend's lane-evaluations are live fragments?  (CPU analysis, oracle raster records of a config's frame.)
For a sample of the visible splats: L = pixels with a live fragment (|q| <= 2 on both axes, exp(-|q|^2) a >= 1/255), and the number
of pixel blocks of several shapes that the blend's own separating-axis test (BlockMayTouch) would hand the splat to.  A wave
evaluates every pixel of a surviving block, so lanes per splat = block pixels x surviving blocks; utilisation = L / that.
    python scripts/analyze_lane_utilisation.py [C2] [sample]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from unitygaussiansplatting_amd import camera, creator, scenes

key = sys.argv[1] if len(sys.argv) > 1 else "C2"
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
cfg = scenes.CONFIGS[key]
a = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name=key)
cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.0), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
orc = O.Oracle(a)
P = camera.frame_params(cam, camera.Transform())
orc.calc_view(P)
recs, rects, vis = orc.raster_records(P)
m = np.unpackbits(vis.view(np.uint8), bitorder="little")[:orc.n].astype(bool)
r = np.asarray(recs)[m]
rng = np.random.default_rng(0)
pick = rng.choice(len(r), min(sample, len(r)), replace=False)
r = r[pick]
raw = np.ascontiguousarray(r).view(np.float32).reshape(len(r), 8)
cx, cy, a1x, a1y, a2x, a2y = (raw[:, k].astype(np.float64) for k in range(6))
c1 = np.ascontiguousarray(r).view(np.uint32).reshape(len(r), 8)[:, 7]
alpha = (c1 & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)
inv1 = 1.0 / (a1x * a1x + a1y * a1y); inv2 = 1.0 / (a2x * a2x + a2y * a2y)
u1x, u1y, u2x, u2y = a1x * inv1, a1y * inv1, a2x * inv2, a2y * inv2
r2 = np.log(np.maximum(255.0 * alpha, 1e-30)) * 1.0001 + 1e-3
rr = np.sqrt(np.maximum(r2, 0))
ex = np.minimum(2 * (np.abs(a1x) + np.abs(a2x)), rr * np.sqrt(a1x ** 2 + a2x ** 2)) + 0.02
ey = np.minimum(2 * (np.abs(a1y) + np.abs(a2y)), rr * np.sqrt(a1y ** 2 + a2y ** 2)) + 0.02
W, H = cfg.width, cfg.height
SHAPES = [(16, 16), (8, 8), (8, 4), (4, 4), (8, 2), (4, 2), (2, 2), (1, 1)]
tot = {s: 0 for s in SHAPES}
live = 0
for i in range(len(r)):
    x0, x1 = int(max(0, np.ceil(cx[i] - ex[i] - 0.5))), int(min(W - 1, np.floor(cx[i] + ex[i] - 0.5)))
    y0, y1 = int(max(0, np.ceil(cy[i] - ey[i] - 0.5))), int(min(H - 1, np.floor(cy[i] + ey[i] - 0.5)))
    if x0 > x1 or y0 > y1:
        continue
    xs = np.arange(x0, x1 + 1) + 0.5 - cx[i]; ys = np.arange(y0, y1 + 1) + 0.5 - cy[i]
    dx, dy = np.meshgrid(xs, ys)
    q1 = dy * u1y[i] + dx * u1x[i]; q2 = dy * u2y[i] + dx * u2x[i]
    lv = (np.abs(q1) <= 2) & (np.abs(q2) <= 2) & (q1 * q1 + q2 * q2 <= r2[i] / 1.0001)
    live += int(lv.sum())
    for (bw, bh) in SHAPES:
        bx0, bx1, by0, by1 = x0 // bw, x1 // bw, y0 // bh, y1 // bh
        bx, by = np.meshgrid(np.arange(bx0, bx1 + 1), np.arange(by0, by1 + 1))
        ccx, ccy = bx * bw + bw / 2.0, by * bh + bh / 2.0           # block centre in pixel-centre coordinates: pixels [b*bw+0.5, b*bw+bw-0.5]
        hx, hy = (bw - 1) / 2.0, (bh - 1) / 2.0
        ddx, ddy = ccx - cx[i], ccy - cy[i]
        d1 = np.abs(ddy * u1y[i] + ddx * u1x[i]) - (hx * abs(u1x[i]) + hy * abs(u1y[i]))
        d2 = np.abs(ddy * u2y[i] + ddx * u2x[i]) - (hx * abs(u2x[i]) + hy * abs(u2y[i]))
        m1, m2 = np.maximum(d1, 0), np.maximum(d2, 0)
        ok = (m1 <= 2.001) & (m2 <= 2.001) & (m1 * m1 + m2 * m2 <= r2[i])
        tot[(bw, bh)] += int(ok.sum()) * bw * bh
print(f"{key}: {len(r)} of the visible splats sampled; live fragments per splat {live / len(r):.1f}")
print("block   lane-evaluations per splat   live share")
for s in SHAPES:
    print(f"{s[0]:2d}x{s[1]:<2d}   {tot[s] / len(r):10.1f}                {100.0 * live / tot[s]:5.1f} %")
