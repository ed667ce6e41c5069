"""Generates tests/golden/*.npz.

PROVENANCE: these vectors are produced by THIS repository's oracle (oracle/gs_oracle.cpp), not by the reference:
the reference (HLSL + Unity C#) cannot be built or run offline and ships no golden vectors for this path
(DESIGN.md "parity unpinned").  They freeze the oracle's behaviour so that a later change to the oracle or to the
canonical arithmetic is caught, and give the -m gpu tests a second, file-based expectation.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from unitygaussiansplatting_amd import camera, creator, scenes  # noqa: E402

CASES = {"medium_700": ("Medium", 700, 41), "veryhigh_300": ("VeryHigh", 300, 42), "high_500": ("High", 500, 43)}


def camera_for(name):
    return camera.Camera(position=(1.5, 1.0, 5.0), target=(0.1, 0.0, 0.0), pixelWidth=96, pixelHeight=64, fieldOfView=45.0)


def transform_for(name):
    return camera.Transform(position=(0.1, -0.1, 0.2), rotation=(0.0, 0.1305, 0.0, 0.9914))


def main():
    for name, (quality, n, seed) in CASES.items():
        raw = scenes.make_splats(n, seed, 2.0, logscale_mu=-2.8, logscale_sigma=0.6)
        a = creator.CreateAssetFromSplats(raw, quality, name=name)
        cam, tr = camera_for(name), transform_for(name)
        orc = O.Oracle(a)
        ms = camera.sort_matrix(cam, tr.localToWorldMatrix)
        orc.sort(ms)
        order1 = orc.order.copy()
        keys1 = orc.keys.copy()
        P = camera.frame_params(cam, tr)
        view = orc.calc_view(P).copy()
        rt0 = orc.draw(P, 0)
        pairs = orc.tile_pairs
        rt1 = orc.draw(P, 1)
        r32, r8 = O.resolve(rt0, (0.0, 0.0, 0.0, 1.0))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), quality=quality, n=n,
                            posData=a.posData, otherData=a.otherData, colorData=a.colorData, shData=a.shData,
                            chunkData=a.chunkData if a.chunkData is not None else np.zeros(0, np.uint8),
                            fmt=np.array([int(a.posFormat), int(a.scaleFormat), int(a.colorFormat), int(a.shFormat)]),
                            order=order1, keys=keys1, view=view.view(np.uint32).reshape(-1, 10), rt_exact=rt0, rt_fast=rt1,
                            resolved8=r8, tile_pairs=pairs)
        print(name, "pairs", pairs, "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
