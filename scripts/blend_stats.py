"""Where the blend's wave-instructions go: runs a configuration's frames through a -DGS_BLEND_STATS build of the library (scripts/build_variants.py
stats:GS_BLEND_STATS; GSPLAT_LIB must point at it) and prints the kernel's own counters per frame:

    GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/stats.so python scripts/blend_stats.py C2 [frames] [visible|full]

stagings = wave x batch (every wave of a tile stages NT/NW records per batch); chunks = wave x 64 records tested against the wave's 8x8 quadrant;
bbox = bounding-box hits among them; survivors = records a wave walked (both tests passed); live = fragments blended (lane x record)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import _lib, camera, creator, scenes
from unitygaussiansplatting_amd.asset import GaussianSplatAsset
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget, SortMode

key = sys.argv[1] if len(sys.argv) > 1 else "C2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "visible"
cfg = scenes.CONFIGS[key]
cache = f"/tmp/gsplat_cache/{key}.json"
if os.path.exists(cache):
    asset = GaussianSplatAsset.Load(cache)
else:
    asset = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name=key)
    asset.Save("/tmp/gsplat_cache")
ctx = GpuContext(0)
r = GaussianSplatRenderer(ctx, asset)
if mode == "visible":
    r.sortMode = SortMode.Visible
r.OnEnable()
rt = RenderTarget(ctx, cfg.width, cfg.height)
has_stats = hasattr(_lib.lib(), "gs_debug_blend_stats")
if has_stats:
    fn = _lib.lib().gs_debug_blend_stats
    fn.restype = C.c_int32; fn.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
buf = (C.c_uint64 * 8)()
for f in range(8 + frames):
    cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * f), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
    r.SortPoints(cam); r.CalcViewData(cam); rt.Clear(); r.Draw(cam, rt)
    try:
        st = r.FrameStats()
    except _lib.GsError as e:
        if e.code != -6: raise
        rt.Clear(); r.Draw(cam, rt); st = r.FrameStats()
    if has_stats:
        assert fn(buf, 1) == 0
    if not has_stats or f < 8:
        continue
    stg, chunks, bbox, surv, live, wgs = (int(buf[i]) for i in range(6))
    nw = (st.tile_w // 8) * (st.tile_h // 8)
    print(json.dumps(dict(cfg=key, mode=mode, frame=f, tile=f"{st.tile_w}x{st.tile_h}", P=int(st.tile_pairs), visible=int(st.visible_splats), workgroups_with_a_list=wgs,
                          wave_stagings=stg, batches=stg // nw, records_staged=stg // nw * 64 * nw, wave_chunks_tested=chunks, bbox_hits=bbox, survivors_walked=surv,
                          live_fragments=live, bbox_hit_share_of_lanes=round(bbox / max(1, chunks * 64), 4), survivor_share_of_bbox=round(surv / max(1, bbox), 4),
                          live_share_of_walked_lanes=round(live / max(1, surv * 64), 4))), flush=True)

# the last frame's per-workgroup timeline (dispatch order = heaviest tile first): -DGS_BLEND_TL builds
import numpy as np
if not hasattr(_lib.lib(), "gs_debug_blend_timeline"):
    sys.exit(0)
nw = (st.tile_w // 8) * (st.tile_h // 8)
tl = np.zeros((8192, 8), np.uint64)
fn2 = _lib.lib().gs_debug_blend_timeline
fn2.restype = C.c_int32; fn2.argtypes = [C.c_void_p, C.c_size_t]
assert fn2(tl.ctypes.data_as(C.c_void_p), tl.nbytes) == 0
ntiles = st.tiles_x * st.tiles_y
t = tl[:ntiles].astype(np.int64)
t0 = t[:, 0].min()
s_, e_ = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
dur, bat, chain, lst = e_ - s_, t[:, 2], t[:, 3], t[:, 4]
span = e_.max()
slots = 2048 * 4 // nw
print(f"timeline ({key}, {st.tile_w}x{st.tile_h}, {ntiles} workgroups, {slots} slots): span {span:.1f} us; sum of durations / slots = {dur.sum() / slots:.1f} us; "
      f"duration median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f}")
print("  workgroups running at t = f x span: " + "  ".join(f"{int(100 * f)}%:{int(((s_ <= f * span) & (e_ > f * span)).sum())}" for f in (0.05, 0.2, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95)))
print(f"  start times: p50 {np.median(s_):.1f} p75 {np.percentile(s_, 75):.1f} p90 {np.percentile(s_, 90):.1f} max {s_.max():.1f}")
ns = np.maximum(1, chain)
print(f"  per tile: longest wave chain median {np.median(chain):.0f} max {chain.max()}; duration / chain survivor (tiles with chain >= 200): median {np.median((1000 * dur / ns)[chain >= 200]):.0f} ns "
      f"(first {slots} dispatched: {np.median((1000 * dur / ns)[:slots][chain[:slots] >= 200]):.0f} ns, later: {np.median((1000 * dur / ns)[slots:][chain[slots:] >= 200]) if (chain[slots:] >= 200).any() else 0:.0f} ns)")
xcc = (t[:, 6] >> 32) & 0xf
print("  per XCD: workgroups, sum of chains, last end / span: " + "  ".join(f"{x}: {int((xcc == x).sum())} {int(chain[xcc == x].sum())} {e_[xcc == x].max() / span:.2f}" for x in range(8)))
print("  last to finish: (dispatch index, start, end, list, batches, chain)")
for i in np.argsort(-e_)[:10]:
    print(f"    {i:5d} {s_[i]:7.1f} {e_[i]:7.1f} {lst[i]:6d} {bat[i]:3d} {chain[i]:5d}")
print("  longest: (dispatch index, start, end, list, batches, chain, ns per chain survivor)")
for i in np.argsort(-dur)[:10]:
    print(f"    {i:5d} {s_[i]:7.1f} {e_[i]:7.1f} {lst[i]:6d} {bat[i]:3d} {chain[i]:5d} {1000 * dur[i] / ns[i]:6.0f}")
np.save(os.path.join(ROOT, "gpurun_out", f"blend_timeline_{key}.npy"), tl[:ntiles])
