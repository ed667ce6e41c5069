"""Per-tile timeline of the blend kernel's last launch (library built with -DGS_EXP_BLEND_TIMELINE; GSPLAT_LIB=...)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import _lib, camera, creator, scenes
from unitygaussiansplatting_amd.asset import GaussianSplatAsset
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget
key = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = scenes.CONFIGS[key]
cache = f"/tmp/gsplat_cache/{key}.json"
if os.path.exists(cache):
    asset = GaussianSplatAsset.Load(cache)
else:
    asset = creator.CreateAssetFromSplats(scenes.make_config_splats(cfg), cfg.quality, name=key); asset.Save("/tmp/gsplat_cache")
ctx = GpuContext(0); r = GaussianSplatRenderer(ctx, asset); r.OnEnable()
rt = RenderTarget(ctx, cfg.width, cfg.height)
shape = os.environ.get("TILE")
if shape:
    r.SetTileShape(*[int(v) for v in shape.split("x")])
for f in range(6):
    if f == 5:
        _z = np.zeros(8, np.uint64); C.CDLL(_lib.LIB_PATH).gs_debug_read_blend_stats(_z.ctypes.data_as(C.c_void_p), C.c_int32(1))     # count the last frame only
    cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * f), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
    r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt); r.FrameStats()
lib = C.CDLL(_lib.LIB_PATH)
stw = r.FrameStats()
tiles = stw.tiles_x * stw.tiles_y
nw = (stw.tile_w // 8) * (stw.tile_h // 8)
print(f"tile {stw.tile_w}x{stw.tile_h}: {tiles} tiles of {nw} waves")
stats = np.zeros(8, np.uint64)
lib.gs_debug_read_blend_stats(stats.ctypes.data_as(C.c_void_p), C.c_int32(1))
buf = np.zeros((65536, 8), np.uint64)
assert lib.gs_debug_read_blend_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
t = buf[:tiles].astype(np.int64)
t0 = t[:, 0].min()
st, en, cnt, bat = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2], t[:, 3]
dur = en - st
surv, tstage, tproc = t[:, 4], t[:, 5] / 100.0, t[:, 6] / 100.0
print(f"wave 0 of each tile: survivors total {surv.sum()} (per walked batch {surv.sum() / max(1, bat.sum()):.1f}); time staging+barriers {tstage.sum():.0f} us, processing (incl. waiting at the next barrier for the slower waves: no) {tproc.sum():.0f} us")
print(f"per walked batch: staging {tstage.sum() / max(1, bat.sum()):.2f} us, processing {tproc.sum() / max(1, bat.sum()):.2f} us; per survivor {1000 * tproc.sum() / max(1, surv.sum()):.1f} ns")
print(f"tiles {tiles}: kernel span {en.max():.1f} us; tile duration median {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f} us")
print(f"start times: median {np.median(st):.1f} p90 {np.percentile(st,90):.1f} max {st.max():.1f}")
print(f"list length: median {np.median(cnt):.0f} max {cnt.max()};  batches walked / batches in list: {bat.sum()} / {((cnt + 255) // 256).sum()}  ({100.0 * bat.sum() / max(1, ((cnt + 255) // 256).sum()):.1f} %)")
print(f"sum of tile durations {dur.sum():.0f} us over {2048 * 4 // nw} slots = {dur.sum() / (2048 * 4 // nw):.1f} us")
idx = np.argsort(-en)[:12]
print("last tiles to finish: (tile, start, end, list, batches)")
for i in idx: print(f"  {i:5d} {st[i]:7.1f} {en[i]:7.1f} {cnt[i]:6d} {bat[i]:4d}")
idx = np.argsort(-dur)[:8]
print("longest tiles: (tile, start, end, list, batches, us/batch)")
for i in idx: print(f"  {i:5d} {st[i]:7.1f} {en[i]:7.1f} {cnt[i]:6d} {bat[i]:4d} {dur[i] / max(bat[i],1):6.2f}")
span = en.max()
print("resident tiles over time (of 2048 slots): " + "  ".join(f"{int(100 * f)}%:{int(((st <= f * span) & (en > f * span)).sum())}" for f in (0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)))
print(f"wave activity (last frame): sum of batches walked by each wave {int(stats[0])} / sum over tiles of waves x batches {int(stats[1])} = {float(stats[0]) / max(1.0, float(stats[1])):.3f};  waves by own/tile batches in fifths: {[int(v) for v in stats[2:7]]}")
