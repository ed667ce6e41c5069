"""Stage-time A/B harness for kernel tuning on a GPU box.

    python scripts/bench_stages.py [C2] [frames]         # uses GSPLAT_LIB if set; caches the asset under /tmp/gsplat_cache
Prints one line: per-stage mean ms over `frames` frames (hipEvent ring), P, and the frame wall time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd.asset import GaussianSplatAsset
from unitygaussiansplatting_amd._lib import GsError
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget, SortMode

key = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mode = int(os.environ.get("GS_BLEND_MODE", "0"))
cfg = scenes.CONFIGS[key]
cache = f"/tmp/gsplat_cache/{key}.json"
if os.path.exists(cache):
    asset = GaussianSplatAsset.Load(cache)
else:
    asset = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name=key)
    asset.Save("/tmp/gsplat_cache")
ctx = GpuContext(0)
r = GaussianSplatRenderer(ctx, asset)
r.sortMode = SortMode.Visible if os.environ.get("GS_SORT_MODE", "visible") == "visible" else SortMode.Full      # GS_SORT_MODE=full: the reference-shaped sort
r.OnEnable()
r.blendMode = mode
r.m_SHOrder = int(os.environ.get("GS_SH_ORDER", "3"))
rt = RenderTarget(ctx, cfg.width, cfg.height)
cam_at = lambda f: camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * f), pixelWidth=cfg.width,
                                 pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
prepared = {}
def frame(f):
    if f not in prepared:
        cam = cam_at(f)
        prepared[f] = (r.SortMatrix(cam), r.FrameParams(cam))
    m16, p = prepared[f]
    r.SortPointsPrepared(m16); r.CalcViewDataPrepared(p); rt.Clear(); r.DrawPrepared(p, rt); rt.ResolveAsync((0, 0, 0, 1))
for f in range(5):
    frame(f)
    try:
        r.FrameStats()
    except GsError as e:
        if e.code != -6: raise
        frame(f); r.FrameStats()
st = r.FrameStats()
r.ReservePairs(int(st.tile_pairs * 1.5) + (1 << 20))
for f in range(5, 5 + frames):
    cam = cam_at(f); prepared[f] = (r.SortMatrix(cam), r.FrameParams(cam))
from unitygaussiansplatting_amd import _lib as _L
_L.check(_L.lib().gs_renderer_set_blend_mode(r._r_h, mode), "gs_renderer_set_blend_mode")
noprof = os.environ.get("GS_NOPROF", "0") == "1"
if not noprof:
    r.SetProfiling(frames)
ctx.Synchronize()
t0 = time.perf_counter()
for f in range(5, 5 + frames):
    frame(f)
ctx.Synchronize()
wall = (time.perf_counter() - t0) / frames * 1e3
st = r.FrameStats()
if noprof:
    print(json.dumps(dict(wall_ms=round(wall, 4), P=int(st.tile_pairs), lib="noprof", cfg=key)), flush=True)
    sys.exit(0)
t = r.StageTimes()
out = {k: round(getattr(t, k), 4) for k, _ in t._fields_ if k.endswith("_ms") and k != "resolve_ms"}
out.update(wall_ms=round(wall, 4), P=int(st.tile_pairs), V=int(st.visible_splats), sort=("visible" if r.SortModeActive() else "full"), lib=os.path.basename(os.environ.get("GSPLAT_LIB", "default")), mode=mode, cfg=key, sh=r.m_SHOrder)
print(json.dumps(out), flush=True)
