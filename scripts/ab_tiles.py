"""Tile-shape A/B on a GPU box, one process per configuration (the asset is built once):

    python scripts/ab_tiles.py C2 [frames] [shape ...]      # shapes like 16x16 32x16 32x32 auto (default: the three fixed ones)
For every shape: hipEvent stage means over `frames` profiled frames, then the un-instrumented wall time per frame, one JSON line each.
GSPLAT_LIB selects a variant build (a build without gs_renderer_set_tile_shape is measured once, as it is)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd.asset import GaussianSplatAsset
from unitygaussiansplatting_amd._lib import GsError
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget

key = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
shapes = sys.argv[3:] or ["16x16", "32x16", "32x32"]
cfg = scenes.CONFIGS[key]
cache = f"/tmp/gsplat_cache/{key}.json"
if os.path.exists(cache):
    asset = GaussianSplatAsset.Load(cache)
else:
    asset = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name=key)
    asset.Save("/tmp/gsplat_cache")
ctx = GpuContext(0)
r = GaussianSplatRenderer(ctx, asset)
r.OnEnable()
rt = RenderTarget(ctx, cfg.width, cfg.height)
prepared = {}
def frame(f):
    if f not in prepared:
        cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * f), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
        prepared[f] = (r.SortMatrix(cam), r.FrameParams(cam))
    m16, p = prepared[f]
    r.SortPointsPrepared(m16); r.CalcViewDataPrepared(p); rt.Clear(); r.DrawPrepared(p, rt); rt.ResolveAsync((0, 0, 0, 1))
has_shapes = hasattr(r, "SetTileShape")
for sh in shapes:
    w, h = (0, 0) if sh == "auto" else tuple(int(v) for v in sh.split("x"))      # auto: the library's own choice (adapts over the warm-up frames)
    if has_shapes:
        try:
            r.SetTileShape(w, h)
        except Exception as e:                                       # an older variant build
            has_shapes = False
    for f in range(8):                                               # warm-up: grows the pair buffers, makes the schedule's cost history, lets the automatic shape settle
        frame(f)
        try:
            st = r.FrameStats()
        except GsError as e:
            if e.code != -6: raise
            frame(f); st = r.FrameStats()
    r.ReservePairs(int(st.tile_pairs * 1.5) + (1 << 20))
    for f in range(6, 6 + frames):
        frame(f)                                                     # builds the prepared structs too
    ctx.Synchronize()
    walls = []
    for rep in range(3):
        t0 = time.perf_counter()
        for f in range(6, 6 + frames):
            frame(f)
        ctx.Synchronize()
        walls.append((time.perf_counter() - t0) / frames * 1e3)
    r.SetProfiling(frames)
    for f in range(6, 6 + frames):
        frame(f)
    ctx.Synchronize()
    st = r.FrameStats()
    t = r.StageTimes()
    r.SetProfiling(0)
    out = {k.replace("_ms", ""): round(getattr(t, k), 4) for k, _ in t._fields_ if k.endswith("_ms") and k not in ("resolve_ms",)}
    out.update(cfg=key, tile=f"{getattr(st, 'tile_w', 16)}x{getattr(st, 'tile_h', 16)}", wall_min=round(min(walls), 4), wall_med=round(sorted(walls)[1], 4),
               P=int(st.tile_pairs), lib=os.path.basename(os.environ.get("GSPLAT_LIB", "default")))
    print(json.dumps(out), flush=True)
    if not has_shapes:
        break
