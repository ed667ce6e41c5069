"""How many (tile, splat) pairs of the rectangle-based binning could an exact tile test drop?  (CPU analysis with the oracle's
per-frame raster records; C2 by default.)  For every visible splat and every tile of its rectangle, the separating-axis test the
blend applies per 8x8 quadrant (BlockMayTouch) is evaluated for the whole 16x16 tile; a pair whose tile fails it can never
produce a fragment.  Prints the share of pairs that survive, by rectangle size.
    python scripts/analyze_pair_culling.py [C2] [tile, e.g. 32x16; default 16x16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from unitygaussiansplatting_amd import camera, creator, scenes

key = sys.argv[1] if len(sys.argv) > 1 else "C2"
TW, TH = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "16x16").split("x"))
cfg = scenes.CONFIGS[key]
a = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name=key)
cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.0), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
orc = O.Oracle(a)
P = camera.frame_params(cam, camera.Transform())
orc.calc_view(P)
recs, rects, vis = orc.raster_records(P)
m = np.unpackbits(vis.view(np.uint8), bitorder="little")[:orc.n].astype(bool)
r = np.asarray(recs)[m]
rc = np.asarray(rects)[m].astype(np.int64)
# SplatRec: cx, cy, a1x, a1y, a2x, a2y, c0, c1 (32 B)
raw = np.frombuffer(np.ascontiguousarray(r).tobytes(), np.float32).reshape(len(r), 8)
cx, cy, a1x, a1y, a2x, a2y = (raw[:, k].astype(np.float64) for k in range(6))
c1 = np.frombuffer(np.ascontiguousarray(r).tobytes(), np.uint32).reshape(len(r), 8)[:, 7]
alpha = (c1 & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)      # color1 = f16 b << 16 | f16 a
inv1 = 1.0 / (a1x * a1x + a1y * a1y); inv2 = 1.0 / (a2x * a2x + a2y * a2y)
u1x, u1y, u2x, u2y = a1x * inv1, a1y * inv1, a2x * inv2, a2y * inv2
r2 = np.log(np.maximum(255.0 * alpha, 1e-30)) * 1.0001 + 1e-3
# pixel rectangles {x0 | y0 << 16, (x1 + 1) | (y1 + 1) << 16} -> tiles of TW x TH
px0, py0, px1, py1 = rc[:, 0] & 0xffff, rc[:, 0] >> 16, (rc[:, 1] & 0xffff) - 1, (rc[:, 1] >> 16) - 1
tx0, ty0 = px0 // TW, py0 // TH
tw, th = px1 // TW - tx0 + 1, py1 // TH - ty0 + 1
total = int((tw * th).sum())
kept = 0
by_size = {}
maxw, maxh = int(tw.max()), int(th.max())
for dy in range(maxh):
    for dx in range(maxw):
        sel = (dx < tw) & (dy < th)
        if not sel.any():
            continue
        bx = (tx0[sel] + dx) * TW + TW / 2.0; by = (ty0[sel] + dy) * TH + TH / 2.0            # block of pixel centres [b - (T - 1) / 2, b + (T - 1) / 2]
        hx, hy = (TW - 1) / 2.0, (TH - 1) / 2.0
        ddx, ddy = bx - cx[sel], by - cy[sel]
        d1 = np.abs(ddy * u1y[sel] + ddx * u1x[sel]) - (hx * np.abs(u1x[sel]) + hy * np.abs(u1y[sel]))
        d2 = np.abs(ddy * u2y[sel] + ddx * u2x[sel]) - (hx * np.abs(u2x[sel]) + hy * np.abs(u2y[sel]))
        m1, m2 = np.maximum(d1, 0), np.maximum(d2, 0)
        ok = (m1 <= 2.001) & (m2 <= 2.001) & (m1 * m1 + m2 * m2 <= r2[sel])
        kept += int(ok.sum())
        sz = (tw[sel] * th[sel])
        for s in np.unique(np.minimum(sz, 26)):
            q = np.minimum(sz, 26) == s
            t, k = by_size.get(int(s), (0, 0))
            by_size[int(s)] = (t + int(q.sum()), k + int(ok[q].sum()))
print(f"{key} at {TW}x{TH} tiles: visible splats {len(r)}, pairs from rectangles {total}, pairs whose tile can hold a fragment {kept} ({100.0 * kept / total:.1f} %)")
print("rect tiles : pairs, surviving share")
for s in sorted(by_size):
    t, k = by_size[s]
    print(f"  {s if s < 26 else '>=26':>5} : {t:9d}  {100.0 * k / t:5.1f} %")
