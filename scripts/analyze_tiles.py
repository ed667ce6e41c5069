"""Tile-list statistics of a config's first camera (CPU, oracle + host-compiled footprint code)."""
import ctypes as C, os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from unitygaussiansplatting_amd import scenes, creator, camera
from unitygaussiansplatting_amd.asset import GaussianSplatAsset
key = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = scenes.CONFIGS[key]
cache = f"/tmp/c2cache/{key}.json"
if os.path.exists(cache):
    a = GaussianSplatAsset.Load(cache)
else:
    a = creator.CreateAssetFromSplats(scenes.make_config_splats(cfg), cfg.quality, name=key); a.Save("/tmp/c2cache")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", "/tmp/libhm.so", os.path.join(ROOT, "tests", "host_math_harness.cpp")])
hm = C.CDLL("/tmp/libhm.so")
cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.0), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
tr = camera.Transform()
orc = O.Oracle(a)
P = camera.frame_params(cam, tr)
v = orc.calc_view(P)
n = a.splatCount
out = np.zeros((n, 5), np.int32); cxy = np.zeros((n, 2), np.float32)
hm.hm_prepare(v.ctypes.data_as(C.c_void_p), C.c_uint32(n), C.byref(P), out.ctypes.data_as(C.c_void_p), cxy.ctypes.data_as(C.c_void_p))
w = np.maximum(out[:, 1] - out[:, 0] + 1, 0); h = np.maximum(out[:, 3] - out[:, 2] + 1, 0)
cnt = (w * h).astype(np.int64)
vis = cnt > 0
print("N", n, "visible", vis.sum(), "pairs", cnt.sum(), "w>0 frac", (v['pos'][:,3] > 0).mean())
print("tiles per visible splat: mean %.2f median %d p90 %d p99 %d max %d" % (cnt[vis].mean(), np.median(cnt[vis]), np.percentile(cnt[vis], 90), np.percentile(cnt[vis], 99), cnt.max()))
tx, ty = (cfg.width + 15) // 16, (cfg.height + 15) // 16
per_tile = np.zeros(tx * ty, np.int64)
for i in np.flatnonzero(vis):
    per_tile.reshape(ty, tx)[out[i, 2]:out[i, 3] + 1, out[i, 0]:out[i, 1] + 1] += 1
print("tiles", tx * ty, "list length: mean %.0f median %.0f p90 %.0f p99 %.0f max %d" % (per_tile.mean(), np.median(per_tile), np.percentile(per_tile, 90), np.percentile(per_tile, 99), per_tile.max()))
ax = np.sqrt(v['axis1'][:, 0] ** 2 + v['axis1'][:, 1] ** 2)[vis]
print("major axis px: median %.2f p90 %.2f p99 %.2f" % (np.median(ax), np.percentile(ax, 90), np.percentile(ax, 99)))
np.save("/tmp/c2cache/per_tile.npy", per_tile)
