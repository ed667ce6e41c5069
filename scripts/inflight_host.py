"""Is the frames-in-flight loop host-bound?  Per lane count: time to ENQUEUE K frames (before the final synchronize) and total time."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import camera, creator, scenes, _lib
from unitygaussiansplatting_amd._lib import check
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext, RenderTarget, SortMode
key = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg = scenes.CONFIGS[key]
asset = creator.CreateAssetFromSplatsNative(scenes.make_config_splats(cfg), cfg.quality, name=key)
W, H = cfg.width, cfg.height
lanes = []
for k in range(3):
    ctx = GpuContext(0); r = GaussianSplatRenderer(ctx, asset); r.sortMode = SortMode.Visible
    if k == 0: r.OnEnable()
    else: r.ShareResourcesOf(lanes[0][1])
    lanes.append((ctx, r, RenderTarget(ctx, W, H)))
prep = []
for i in range(80):
    cam = camera.Camera(position=scenes.orbit_eye(cfg.eye_radius, cfg.eye_elev_deg, 0.25 * i), pixelWidth=W, pixelHeight=H, fieldOfView=cfg.fov_y)
    prep.append((lanes[0][1].SortMatrix(cam), lanes[0][1].FrameParams(cam)))
bg = np.asarray((0, 0, 0, 1), np.float32); bgp = bg.ctypes.data_as(C.POINTER(C.c_float)); lib = _lib.lib()
def frame(i, act):
    m16, p = prep[i]
    for (_, r, _) in act: r.SortPointsPrepared(m16)
    ctx, r, t = act[i % len(act)]
    r.CalcViewDataPrepared(p); t.Clear(); r.DrawPrepared(p, t); check(lib.gs_target_resolve(t._h, bgp, None, None), "resolve")
for nl in (1, 2, 3):
    act = lanes[:nl]
    for i in range(10): frame(i, act)
    for (c, r, _) in act:
        c.Synchronize(); st = r.FrameStats(); r.ReservePairs(int(st.tile_pairs * 1.5) + (1 << 20))
    for rep in range(3):
        for (c, _, _) in act: c.Synchronize()
        t0 = time.perf_counter()
        for i in range(10, 70): frame(i, act)
        t1 = time.perf_counter()
        for (c, _, _) in act: c.Synchronize()
        t2 = time.perf_counter()
        print(f"{key} lanes {nl}: enqueue {(t1 - t0) / 60 * 1e3:.4f} ms/frame, total {(t2 - t0) / 60 * 1e3:.4f} ms/frame", flush=True)
