"""The reference's validation path (GaussianSplatValidator.cs, cameras.json import, ActivateCamera) -- ready for the INRIA models;
tested here on a synthetic scene with a synthetic cameras.json and a reference PNG rendered by the oracle."""
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from common import small_asset
from unitygaussiansplatting_amd import camera, creator, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import validate_refimages as V  # noqa: E402


def colmap_cameras(n=3):
    """cameras.json entries the way the INRIA trainer writes them: position = camera centre, rotation = camera-to-world with
    COLMAP axes (x right, y down, z forward) as columns."""
    cams = []
    for k in range(n):
        eye = np.array(scenes.orbit_eye(5.0, 10.0 + 5 * k, 40.0 * k), np.float64)
        z = -eye / np.linalg.norm(eye)
        x = np.cross(z, [0.0, 1.0, 0.0]); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z], axis=1)
        cams.append(dict(id=k, img_name=f"im{k}", width=160, height=100, position=[float(v) for v in eye], rotation=[[float(v) for v in row] for row in R], fx=200.0, fy=200.0))
    return cams


def test_cameras_json_import_follows_the_reference(tmp_path):
    d = tmp_path / "model" / "point_cloud" / "iteration_30000"
    d.mkdir(parents=True)
    cams = colmap_cameras()
    (tmp_path / "model" / "cameras.json").write_text(json.dumps(cams))               # two directories above the .ply
    got = creator.LoadJsonCamerasFile(str(d / "point_cloud.ply"), True)
    assert len(got) == 3 and creator.LoadJsonCamerasFile(str(d / "point_cloud.ply"), False) is None
    for ci, jc in zip(got, cams):
        R = np.asarray(jc["rotation"], np.float32)
        assert np.allclose(ci.pos, jc["position"]) and ci.fov == 25.0
        assert np.array_equal(np.float32(ci.axisX), R[:, 0]) and np.array_equal(np.float32(ci.axisY), -R[:, 1]) and np.array_equal(np.float32(ci.axisZ), -R[:, 2])
    assert creator.LoadJsonCamerasFile(str(tmp_path / "elsewhere" / "x.ply"), True) is None


def test_activate_camera_is_the_unity_transform():
    """ActivateCamera (GaussianSplatRenderer.cs:660-680): world position through the renderer's localToWorld (mirror scale included),
    world rotation = parent rotation * (local rotation conjugated by the signs of the parent's scale).  Checked from the outside:
    a cameras.json camera that looks at the object-space origin (COLMAP axes: x right, y down, z forward) must, under the sample
    scene's mirrored transform, see the object's origin at the image centre, a point to ITS right on the right and a point ABOVE
    it (COLMAP -y) on top -- i.e. the picture the trainer's camera saw, not its mirror image."""
    from unitygaussiansplatting_amd.asset import CameraInfo
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer
    import json as _json
    a = small_asset(500, 3, "Medium")
    jc = colmap_cameras(3)[1]
    R = np.asarray(jc["rotation"], np.float64)                           # columns: camera x (right), y (down), z (forward) in object space
    a.cameras = [CameraInfo(pos=tuple(jc["position"]), axisX=tuple(R[:, 0]), axisY=tuple(-R[:, 1]), axisZ=tuple(-R[:, 2]), fov=25.0)]
    tr = camera.Transform(position=(0.3, 0.2, -0.1), rotation=V.SCENE_ROTATION, scale=V.SCENE_SCALE)
    r = GaussianSplatRenderer.__new__(GaussianSplatRenderer)
    r.m_Asset, r.transform = a, tr
    cam = camera.Camera(pixelWidth=64, pixelHeight=48)
    GaussianSplatRenderer.ActivateCamera(r, 0, cam)
    o2w = tr.localToWorldMatrix.astype(np.float64)
    assert np.allclose(cam.position, (o2w @ np.append(jc["position"], 1.0))[:3], atol=1e-6)
    Rw = np.asarray(cam.rotation)
    assert np.allclose(Rw.T @ Rw, np.eye(3), atol=1e-6) and np.linalg.det(Rw) > 0
    VP = (cam.projectionMatrix.astype(np.float64) @ cam.worldToCameraMatrix.astype(np.float64))

    def ndc(p_obj):
        c = VP @ (o2w @ np.append(p_obj, 1.0))
        return c[:2] / c[3], c[3]
    centre, w = ndc(np.zeros(3))
    assert w > 0 and np.allclose(centre, 0.0, atol=1e-5)                   # looks at the origin, which is in front
    right, _ = ndc(0.2 * R[:, 0])
    above, _ = ndc(-0.2 * R[:, 1])
    assert right[0] > 0.01 and abs(right[1]) < 1e-5                        # the camera's right is the picture's right (clip x grows to the right)
    assert above[1] > 0.01 and abs(above[0]) < 1e-5                        # COLMAP -y is up (clip y grows upwards)


def test_diff_images_metric_is_the_validators():
    ref = np.zeros((4, 5, 3), np.uint8)
    got = ref.copy()
    got[0, 0] = (2, 0, 0); got[1, 1] = (0, 3, 0); got[2, 2] = (10, 10, 10)
    rmse, psnr, count, dif = V.diff_images(ref, got)
    want_ms = (4 + 9 + 300) / (4 * 5 * 3)
    assert abs(rmse - np.sqrt(want_ms)) < 1e-12 and abs(psnr - (20 * np.log10(255.0) - 10 * np.log10(want_ms))) < 1e-9
    assert count == 2 and tuple(dif[2, 2]) == (50, 50, 50) and tuple(dif[0, 0]) == (10, 0, 0)      # 2/255 is below the 3/255 threshold
    assert V.verdict(95.0, 50) and not V.verdict(95.0, 51) and not V.verdict(89.9, 0)
    assert V.diff_images(ref, ref)[1] == float("inf")


@pytest.mark.gpu
def test_validator_end_to_end_on_a_synthetic_scene(tmp_path):
    """PLY + cameras.json on disk -> scripts/validate_refimages.py -> verdict, against a PNG rendered by the ORACLE from the same
    files with the same camera: 'matches' (PSNR >= 90 dB, <= 50 pixels off by >= 3/255); against a shifted PNG: 'differs'."""
    from PIL import Image
    d = tmp_path / "scene" / "point_cloud" / "iteration_30000"
    d.mkdir(parents=True)
    raw = scenes.make_splats(20000, 5, 3.0)
    ply = str(d / "point_cloud.ply")
    creator.WritePLY(ply, raw)
    (tmp_path / "scene" / "cameras.json").write_text(json.dumps(colmap_cameras()))
    asset = creator.CreateAsset(ply, "Medium")
    assert len(asset.cameras) == 3
    W, H, fov, idx = 160, 100, 45.0, 1
    # the oracle's frame for camera 1 (same transform / camera construction as the script)
    from unitygaussiansplatting_amd.renderer import GaussianSplatRenderer
    tr = camera.Transform(rotation=V.SCENE_ROTATION, scale=V.SCENE_SCALE)
    r = GaussianSplatRenderer.__new__(GaussianSplatRenderer)
    r.m_Asset, r.transform = asset, tr
    cam = camera.Camera(pixelWidth=W, pixelHeight=H, fieldOfView=fov)
    GaussianSplatRenderer.ActivateCamera(r, idx, cam)
    orc = O.Oracle(asset)
    orc.sort(camera.sort_matrix(cam, tr.localToWorldMatrix))
    P = camera.frame_params(cam, tr)
    orc.calc_view(P)
    _, o8 = O.resolve(orc.draw(P, 0), (0.0, 0.0, 0.0, 0.0))
    assert o8[..., :3].mean() > 2.0                                            # the camera sees the scene
    png = str(tmp_path / "ref.png")
    Image.fromarray(o8[..., :3]).save(png)
    assert V.main([ply, str(idx), png, "--fov", str(fov)]) == 0
    Image.fromarray(np.roll(o8[..., :3], 7, axis=1)).save(png)
    assert V.main([ply, str(idx), png, "--fov", str(fov)]) == 1
