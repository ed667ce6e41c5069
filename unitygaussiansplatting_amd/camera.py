"""Minimal stand-ins for the Unity objects the renderer reads per frame: Transform and Camera.

Only what GaussianSplatRenderer.CalcViewData / SortPoints consume (GaussianSplatRenderer.cs:579-639):
cam.worldToCameraMatrix (GL convention, camera looks down -Z), cam.projectionMatrix, pixelWidth/Height,
cam.transform.position, near/far clip planes; transform.localToWorldMatrix / worldToLocalMatrix.
All matrices are float32, row-major, column vectors; products are evaluated in float32 without FMA,
left to right, like UnityEngine.Matrix4x4.operator* on the CPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from ._abi import gs_frame_params

f32 = np.float32


def mat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Matrix4x4.operator*: res[i][j] = a[i][0]*b[0][j] + a[i][1]*b[1][j] + ... in float32, no FMA, left to right.
    (Four rank-1 float32 products added in order: every element sees exactly the scalar sequence of roundings.)"""
    a = np.asarray(a, f32); b = np.asarray(b, f32)
    acc = a[:, 0:1] * b[0:1, :]
    for k in range(1, 4):
        acc = acc + a[:, k:k + 1] * b[k:k + 1, :]
    return acc.astype(f32)


def quat_to_mat3(q: Sequence[float]) -> np.ndarray:
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], np.float64)


@dataclass
class Transform:
    position: Sequence[float] = (0.0, 0.0, 0.0)
    rotation: Sequence[float] = (0.0, 0.0, 0.0, 1.0)     # quaternion xyzw
    scale: Sequence[float] = (1.0, 1.0, 1.0)

    @property
    def localToWorldMatrix(self) -> np.ndarray:
        m = np.eye(4)
        m[:3, :3] = quat_to_mat3(self.rotation) @ np.diag(np.asarray(self.scale, np.float64))
        m[:3, 3] = self.position
        return m.astype(f32)

    @property
    def worldToLocalMatrix(self) -> np.ndarray:
        return np.linalg.inv(self.localToWorldMatrix.astype(np.float64)).astype(f32)


def quat_mul(a: Sequence[float], b: Sequence[float]) -> np.ndarray:
    """Quaternion product a * b (xyzw), i.e. the rotation b followed by a."""
    ax, ay, az, aw = [float(v) for v in a]
    bx, by, bz, bw = [float(v) for v in b]
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], np.float64)


def look_rotation(forward: Sequence[float], up: Sequence[float]) -> np.ndarray:
    """Quaternion.LookRotation as a 3x3 matrix whose columns are the rotated X, Y, Z axes: Z along `forward`,
    X along cross(up, forward), Y = cross(Z, X)."""
    z = np.asarray(forward, np.float64); z = z / np.linalg.norm(z)
    x = np.cross(np.asarray(up, np.float64), z); x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=1)


@dataclass
class Camera:
    """Perspective camera.  `LookAt` builds the pose; Unity's camera-space looks down -Z in worldToCameraMatrix.
    With `rotation` set (3x3, columns = the transform's right / up / forward axes in Unity's convention, e.g. from
    GaussianSplatRenderer.ActivateCamera) the pose is the Unity transform (position, rotation) and target / up are ignored."""
    position: Sequence[float] = (0.0, 0.0, 6.0)
    target: Sequence[float] = (0.0, 0.0, 0.0)
    up: Sequence[float] = (0.0, 1.0, 0.0)
    fieldOfView: float = 39.0965      # vertical, degrees (GSTestScene.unity:277)
    pixelWidth: int = 640
    pixelHeight: int = 360
    nearClipPlane: float = 0.3
    farClipPlane: float = 1000.0
    rotation: Optional[np.ndarray] = None

    @property
    def aspect(self) -> float:
        return self.pixelWidth / self.pixelHeight

    @property
    def worldToCameraMatrix(self) -> np.ndarray:
        eye = np.asarray(self.position, np.float64)
        if self.rotation is not None:
            # Camera.worldToCameraMatrix = scale(1, 1, -1) * transform.worldToLocalMatrix (the camera looks down -Z, OpenGL style)
            R = np.asarray(self.rotation, np.float64)
            m = np.eye(4)
            m[:3, :3] = R.T
            m[:3, 3] = -R.T @ eye
            m[2, :] *= -1.0
            return m.astype(f32)
        fwd = np.asarray(self.target, np.float64) - eye
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.asarray(self.up, np.float64))
        right /= np.linalg.norm(right)
        upv = np.cross(right, fwd)
        m = np.eye(4)
        m[0, :3], m[1, :3], m[2, :3] = right, upv, -fwd
        m[:3, 3] = -m[:3, :3] @ eye
        return m.astype(f32)

    @property
    def projectionMatrix(self) -> np.ndarray:
        t = math.tan(math.radians(self.fieldOfView) * 0.5)
        n, f = self.nearClipPlane, self.farClipPlane
        m = np.zeros((4, 4))
        m[0, 0] = 1.0 / (self.aspect * t)
        m[1, 1] = 1.0 / t
        m[2, 2] = -(f + n) / (f - n)
        m[2, 3] = -2.0 * f * n / (f - n)
        m[3, 2] = -1.0
        return m.astype(f32)


def sort_matrix(cam: Camera, localToWorld: np.ndarray) -> np.ndarray:
    """SortPoints (GaussianSplatRenderer.cs:617-629): worldToCameraMatrix with m20,m21,m22 negated, times model."""
    w2c = cam.worldToCameraMatrix.copy()
    w2c[2, 0] *= f32(-1); w2c[2, 1] *= f32(-1); w2c[2, 2] *= f32(-1)
    return mat_mul(w2c, localToWorld)


def frame_params(cam: Camera, tr: Transform, splatScale: float = 1.0, opacityScale: float = 1.0, shOrder: int = 3,
                 shOnly: bool = False) -> gs_frame_params:
    """CalcViewData's constants (GaussianSplatRenderer.cs:585-606) + UNITY_MATRIX_VP / UNITY_MATRIX_P."""
    p = gs_frame_params()
    matView = cam.worldToCameraMatrix
    matO2W = tr.localToWorldMatrix
    matW2O = tr.worldToLocalMatrix
    proj = cam.projectionMatrix
    put = lambda dst, m: dst.__setitem__(slice(0, 16), [float(v) for v in np.asarray(m, f32).reshape(-1)])
    put(p.matrix_mv, mat_mul(matView, matO2W))
    put(p.matrix_object_to_world, matO2W)
    put(p.matrix_world_to_object, matW2O)
    put(p.matrix_vp, mat_mul(proj, matView))
    p.proj_m00, p.proj_m11 = float(proj[0, 0]), float(proj[1, 1])
    p.screen_w, p.screen_h = float(cam.pixelWidth), float(cam.pixelHeight)
    p.cam_pos_world[0:3] = [float(v) for v in cam.position]
    p.splat_scale, p.opacity_scale = float(splatScale), float(opacityScale)
    p.sh_order, p.sh_only = int(shOrder), int(bool(shOnly))
    p.near_clip, p.far_clip = float(cam.nearClipPlane), float(cam.farClipPlane)
    return p
