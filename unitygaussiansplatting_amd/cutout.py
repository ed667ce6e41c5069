"""GaussianCutout (/root/reference/package/Runtime/GaussianCutout.cs): an ellipsoid or box volume, placed by its own
transform, that removes the splats inside it (or, inverted, outside it) in CSCalcViewData."""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import IntEnum
from typing import Optional, Sequence

import numpy as np

from ._abi import gs_cutout
from .camera import Transform, mat_mul


class Type(IntEnum):            # GaussianCutout.cs:11-15
    Ellipsoid = 0
    Box = 1


@dataclass
class GaussianCutout:
    m_Type: Type = Type.Ellipsoid
    m_Invert: bool = False
    transform: Transform = field(default_factory=Transform)
    isActiveAndEnabled: bool = True

    @staticmethod
    def GetShaderData(self: Optional["GaussianCutout"], rendererMatrix: np.ndarray) -> gs_cutout:
        """GaussianCutout.cs:24-40: matrix = cutout.worldToLocal * renderer.localToWorld; a null/disabled cutout gets
        typeAndFlags = ~0 and is skipped by the kernel."""
        sd = gs_cutout()
        if self is not None and self.isActiveAndEnabled:
            m = mat_mul(self.transform.worldToLocalMatrix, rendererMatrix)
            sd.matrix[0:16] = [float(v) for v in m.reshape(-1)]
            sd.type_and_flags = int(self.m_Type) | (0x100 if self.m_Invert else 0)
        else:
            sd.type_and_flags = 0xFFFFFFFF
        return sd


def shader_data_array(cutouts: Optional[Sequence[Optional[GaussianCutout]]], rendererMatrix: np.ndarray):
    """UpdateCutoutsBuffer (GaussianSplatRenderer.cs:742-764): the _SplatCutouts array for a renderer."""
    n = len(cutouts) if cutouts else 0
    arr = (gs_cutout * max(n, 1))()
    for i in range(n):
        arr[i] = GaussianCutout.GetShaderData(cutouts[i], rendererMatrix)
    return arr, n
