"""Single process: the C2 orbit seen from view 1 (azimuth 45 deg + 0.25 deg per frame) in GS_SORT_FULL, GS_SORT_VISIBLE and with two renderers in flight;
FrameStats after every frame (raises on any sort / binning error).  (What rank 1 of a 2-rank bench renders.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd._lib import GsError
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget, SortMode
cfg = scenes.CONFIGS["C2"]
asset = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name="C2")
W, H = cfg.width, cfg.height
ctx = GpuContext(0); r = GaussianSplatRenderer(ctx, asset); r.OnEnable(); rt = RenderTarget(ctx, W, H)
ctx2 = GpuContext(0); r2 = GaussianSplatRenderer(ctx2, asset); r2.sortMode = SortMode.Visible; r2.ShareResourcesOf(r); rt2 = RenderTarget(ctx2, W, H)
cam = lambda i: camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 45.0 + 0.25 * i), pixelWidth=W, pixelHeight=H, fieldOfView=cfg.fov_y)
for mode in ("full", "visible", "in_flight"):
    r.SetSortMode(SortMode.Full); r.ResetOrder(); r.SetSortMode(SortMode.Full if mode == "full" else SortMode.Visible)
    r2.SetSortMode(SortMode.Full); r2.ResetOrder(); r2.SetSortMode(SortMode.Visible)
    t0 = time.perf_counter()
    for i in range(40):
        c = cam(i)
        lanes = [(r, rt), (r2, rt2)] if mode == "in_flight" else [(r, rt)]
        for (x, _) in lanes: x.SortPoints(c)
        x, t = lanes[i % len(lanes)]
        x.CalcViewData(c); t.Clear(); x.Draw(c, t)
        try:
            st = x.FrameStats()
        except GsError as e:
            if e.code != -6: raise
            x.CalcViewData(c); t.Clear(); x.Draw(c, t); st = x.FrameStats()
    print(mode, "ok", int(st.tile_pairs), int(st.visible_splats), f"{time.perf_counter() - t0:.2f}s", flush=True)
