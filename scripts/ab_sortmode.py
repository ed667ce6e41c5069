"""GS_SORT_FULL against GS_SORT_VISIBLE on a GPU box, same process, same asset, same frames, alternating:

    python scripts/ab_sortmode.py C2 [frames] [reps]
Per mode and repetition one JSON line: hipEvent stage means over `frames` profiled frames and the un-instrumented wall time per frame
(min / median of three regions).  The two renderers share the context; each is warmed up (pair buffers grown, tile schedule history,
automatic tile shape settled) before it is measured.  GSPLAT_LIB selects a variant build."""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd.asset import GaussianSplatAsset
from unitygaussiansplatting_amd._lib import GsError
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget, SortMode

key = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = scenes.CONFIGS[key]
cache = f"/tmp/gsplat_cache/{key}.json"
if os.path.exists(cache):
    asset = GaussianSplatAsset.Load(cache)
else:
    asset = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name=key)
    asset.Save("/tmp/gsplat_cache")
ctx = GpuContext(0)
rt = RenderTarget(ctx, cfg.width, cfg.height)
prepared = {}


def frame(r, f):
    if f not in prepared:
        cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * f), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
        prepared[f] = (r.SortMatrix(cam), r.FrameParams(cam))
    m16, p = prepared[f]
    r.SortPointsPrepared(m16); r.CalcViewDataPrepared(p); rt.Clear(); r.DrawPrepared(p, rt); rt.ResolveAsync((0, 0, 0, 1))


renderers = {}
for name, mode in (("full", SortMode.Full), ("visible", SortMode.Visible)):
    r = GaussianSplatRenderer(ctx, asset)
    r.sortMode = mode
    r.OnEnable()
    for f in range(8):
        frame(r, f)
        try:
            st = r.FrameStats()
        except GsError as e:
            if e.code != -6: raise
            frame(r, f); st = r.FrameStats()
    r.ReservePairs(int(st.tile_pairs * 1.5) + (1 << 20))
    renderers[name] = r

for rep in range(reps):
    for name, r in renderers.items():
        for f in range(6, 6 + frames):
            frame(r, f)
        ctx.Synchronize()
        walls = []
        for _ in range(3):
            t0 = time.perf_counter()
            for f in range(6, 6 + frames):
                frame(r, f)
            ctx.Synchronize()
            walls.append((time.perf_counter() - t0) / frames * 1e3)
        r.SetProfiling(frames)
        for f in range(6, 6 + frames):
            frame(r, f)
        ctx.Synchronize()
        st = r.FrameStats()
        crc = zlib.crc32(rt.Download().tobytes())
        t = r.StageTimes()
        r.SetProfiling(0)
        out = {k.replace("_ms", ""): round(getattr(t, k), 4) for k, _ in t._fields_ if k.endswith("_ms") and k not in ("resolve_ms",)}
        out.update(cfg=key, sort_mode=name, active=bool(r.SortModeActive()), tile=f"{st.tile_w}x{st.tile_h}", wall_min=round(min(walls), 4), wall_med=round(sorted(walls)[1], 4),
                   P=int(st.tile_pairs), V=int(st.visible_splats), frame_crc=crc, tie_long_runs=int(getattr(st, 'tie_long_runs', 0)), lib=os.path.basename(os.environ.get("GSPLAT_LIB", "default")))
        print(json.dumps(out), flush=True)
