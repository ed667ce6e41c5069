"""Phase timeline of bin_emit's last launch (needs a library built with -DGS_EXP_BIN_TIMELINE; GSPLAT_LIB=...): renders a few C2
frames and prints, per phase, when partitions reach it (us since the first workgroup started; 100 MHz clock)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import _lib, camera, creator, scenes
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget
key = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = scenes.CONFIGS[key]
from unitygaussiansplatting_amd.asset import GaussianSplatAsset
cache = f"/tmp/gsplat_cache/{key}.json"
if os.path.exists(cache):
    asset = GaussianSplatAsset.Load(cache)
else:
    asset = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name=key); asset.Save("/tmp/gsplat_cache")
ctx = GpuContext(0)
r = GaussianSplatRenderer(ctx, asset); r.OnEnable()
rt = RenderTarget(ctx, cfg.width, cfg.height)
for f in range(6):
    cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * f), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
    r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    try: r.FrameStats()
    except Exception: pass
lib = C.CDLL(_lib.LIB_PATH)
parts = (asset.splatCount + 2047) // 2048
buf = np.zeros((32768, 8), np.uint64)
assert lib.gs_debug_read_bin_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
t = buf[:parts].astype(np.int64)
t0 = t[:, 0].min()
us = (t[:, :6] - t0) / 100.0
names = ["loop top", "ticket", "loads in", "scan done (w0)", "after barrier", "emitted (w0)"]
print(f"{key}: parts={parts}")
print(f"{'phase':16s} {'min':>8s} {'p10':>8s} {'median':>8s} {'p90':>8s} {'max':>8s}   dt median from previous phase")
for i, nm in enumerate(names):
    c = us[:, i]
    dt = np.median(us[:, i] - us[:, i - 1]) if i else 0.0
    print(f"{nm:16s} {c.min():8.2f} {np.percentile(c,10):8.2f} {np.median(c):8.2f} {np.percentile(c,90):8.2f} {c.max():8.2f}   {dt:8.2f}")
d = us[:, 5] - us[:, 0]
print("partition duration: median %.2f p90 %.2f max %.2f" % (np.median(d), np.percentile(d, 90), d.max()))
print("start by partition (every 128th):", np.round(us[::128, 0], 1).tolist())
print("end by partition (every 128th):  ", np.round(us[::128, 5], 1).tolist())
for i, nm in enumerate(names[1:], 1):
    print(f"dt {nm:16s} by start order deciles:", np.round([np.median((us[:, i] - us[:, i - 1])[np.argsort(us[:, 0])][k * parts // 10:(k + 1) * parts // 10]) for k in range(10)], 2).tolist())
