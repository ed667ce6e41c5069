"""First-light check on a GPU box: HIP path vs oracle on a small scene, with stage timings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from unitygaussiansplatting_amd import scenes, creator, camera
from unitygaussiansplatting_amd.renderer import GpuContext, GaussianSplatRenderer, RenderTarget, GpuSorting
import oracle_lib as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 360)
quality = sys.argv[4] if len(sys.argv) > 4 else "Medium"
ctx = GpuContext(0)
print("device:", ctx.DeviceInfo(), flush=True)

# ---- stand-alone sorter ----
rng = np.random.default_rng(0)
for cnt in (1, 255, 4096, 4097, 100000, 1 << 20):
    keys = rng.integers(0, 2**32, cnt, dtype=np.uint64).astype(np.uint32)
    if cnt == 100000: keys &= 0xff00ff
    vals = np.arange(cnt, dtype=np.uint32)
    s = GpuSorting(ctx, cnt)
    k, v = s.DispatchHost(keys, vals)
    ko, vo = O.sort_pairs(keys, vals)
    print("sorter", cnt, "keys ok", np.array_equal(k, ko), "vals ok", np.array_equal(v, vo), flush=True)
    s.Dispose()

cfg = scenes.CONFIGS["C1"]
raw = scenes.make_splats(n, 1, cfg.extent, cfg.surface_frac, cfg.logscale_mu, cfg.logscale_sigma)
asset = creator.CreateAssetFromSplats(raw, quality)
cam = camera.Camera(position=scenes.orbit_eye(6.0, 10.0, 20.0), pixelWidth=W, pixelHeight=H)
r = GaussianSplatRenderer(ctx, asset)
r.OnEnable()
r.SetProfiling(True)
rt = RenderTarget(ctx, W, H)
orc = O.Oracle(asset)
P = r.FrameParams(cam)
ms = camera.sort_matrix(cam, r.transform.localToWorldMatrix)
for frame in range(2):
    r.SortPoints(cam)
    orc.sort(ms)
    order = r.DownloadOrder()
    print("frame", frame, "order bit-exact:", np.array_equal(order, orc.order), flush=True)
r.CalcViewData(cam)
v = r.DownloadView()
vo = orc.calc_view(P)
b0 = v.view(np.uint32).reshape(-1, 10); b1 = vo.view(np.uint32).reshape(-1, 10)
print("view mismatching words:", int((b0 != b1).sum()), "of", b0.size, flush=True)
for mode in (0, 1):
    r.blendMode = mode
    rt.Clear()
    r.Draw(cam, rt)
    st = r.FrameStats()
    img = rt.Download()
    ref = orc.draw(P, mode)
    a = O.f16_to_f32(img); b = O.f16_to_f32(ref)
    d = np.abs(a - b)
    print(f"mode {mode}: pairs gpu {st.tile_pairs} oracle {orc.tile_pairs} visible {st.visible_splats}/{orc.visible} "
          f"max-abs {d.max():.3e} mean-abs {d.mean():.3e} exact-equal px {(img == ref).all(axis=2).mean():.4f}", flush=True)
t = r.StageTimes()
print({k: round(getattr(t, k), 4) for k, _ in t._fields_})
o32, o8 = rt.Resolve((0, 0, 0, 0))
r32, r8 = O.resolve(ref)
print("resolve max diff 8-bit:", int(np.abs(o8.astype(int) - r8.astype(int)).max()), "f32:", float(np.abs(o32 - r32).max()))
