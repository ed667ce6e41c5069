"""Review item 4 of round 5 (one bounded experiment): the blend's heaviest tiles -- the first K % of the schedule -- composited by two
workgroups each, every wave owning an 8x4 half of its quadrant (scripts/variants/r05_blend_split.patch; the variant library reads
GSPLAT_BLEND_SPLIT_PCT at every draw, so one process sweeps K on the same asset, context and frames, alternating):

    GSPLAT_LIB=unitygaussiansplatting_amd/variants/r05_split4.so python scripts/ab_blend_split.py C2 30 0,3,8 2
One JSON line per (K, repetition): wall ms / frame (median of three un-instrumented regions), the blend stage by hipEvents, the CRC of the
last frame's RGBA16F target (the split must not change a bit).  The shipped library ignores the variable (a control for the sweep's noise)."""
import json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd.asset import GaussianSplatAsset
from unitygaussiansplatting_amd._lib import GsError
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget, SortMode

key = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sweep = [s for s in (sys.argv[3] if len(sys.argv) > 3 else "0,3,8").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
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


r = GaussianSplatRenderer(ctx, asset)
r.sortMode = SortMode.Visible
r.OnEnable()
for f in range(8):
    frame(r, f)
    try:
        st = r.FrameStats()
    except GsError as e:
        if e.code != -6: raise
        frame(r, f); st = r.FrameStats()
r.ReservePairs(int(st.tile_pairs * 1.5) + (1 << 20))

for rep in range(reps):
    for k in sweep:
        os.environ["GSPLAT_BLEND_SPLIT_PCT"] = k
        for f in range(6, 6 + frames):                     # the schedule (last frame's tile costs) settles under this K
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
        t = r.StageTimes()
        r.SetProfiling(0)
        crc = zlib.crc32(rt.Download().tobytes())
        print(json.dumps(dict(cfg=key, lib=os.path.basename(os.environ.get("GSPLAT_LIB", "default")), split_pct=float(k), rep=rep, tile=f"{st.tile_w}x{st.tile_h}",
                              wall_min=round(min(walls), 4), wall_med=round(sorted(walls)[1], 4), blend_us=round(t.blend_ms * 1e3, 1), pair_sort_us=round(t.pair_sort_ms * 1e3, 1),
                              P=int(st.tile_pairs), frame_crc=f"{crc:08x}")), flush=True)
