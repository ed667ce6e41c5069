"""GPU box: where do the HIP view records differ from oracle/_ref's fused build / from the oracle?  (debugging aid for tests/test_gpu_ref.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O, ref_lib as R
from unitygaussiansplatting_amd import camera, creator, scenes
from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer, GpuContext
key = sys.argv[1] if len(sys.argv) > 1 else "C1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = scenes.CONFIGS[key]
a = creator.CreateAssetFromSplats(scenes.make_config_splats(cfg, n), cfg.quality, name=key)
ctx = GpuContext(0)
r = GaussianSplatRenderer(ctx, a); r.OnEnable()
ref = R.Ref(a, "fused"); orc = O.Oracle(a)
print("ref compiler:", R.fused_compiler())
for az, radius in [(0.0, cfg.eye_radius), (90.25, cfg.eye_radius), (200.0, 0.15 * cfg.eye_radius)]:
    cam = camera.Camera(position=scenes.orbit_eye(radius, cfg.eye_elev_deg, az), pixelWidth=cfg.width, pixelHeight=cfg.height, fieldOfView=cfg.fov_y)
    P = r.FrameParams(cam)
    r.CalcViewData(cam)
    g = r.DownloadView().view(np.uint32).reshape(-1, 10)
    o = orc.calc_view(P).view(np.uint32).reshape(-1, 10).copy()
    f = ref.calc_view(P).view(np.uint32).reshape(-1, 10).copy()
    for nm, x, y in (("gpu-vs-oracle", g, o), ("gpu-vs-ref", g, f), ("oracle-vs-ref", o, f)):
        d = x != y
        rows = np.flatnonzero(d.any(1))
        print(f"az {az}: {nm}: {len(rows)} records differ; per field {d.sum(0).tolist()}")
        for i in rows[:3]:
            print("   ", i, [hex(v) for v in x[i]], [hex(v) for v in y[i]], "floats", x[i, :8].view(np.float32), y[i, :8].view(np.float32))
