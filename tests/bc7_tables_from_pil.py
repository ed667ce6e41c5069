"""How the BC7 partition / anchor tables of unitygaussiansplatting_amd/bc7.py were obtained -- and a check that they still
match: probe an INDEPENDENT BC7 decoder (Pillow's BcnDecode, reached through an in-memory DX10 DDS file) with blocks whose
decoded colours reveal the subset of every texel and the anchor texels.

    python tests/bc7_tables_from_pil.py          # prints the tables, compares with bc7.py
"""
import io
import struct

import numpy as np


def dds_bc7(blocks: bytes, w: int, h: int) -> bytes:
    hdr = struct.pack("<4sIIIIIII44x", b"DDS ", 124, 0x1007 | 0x80000, h, w, len(blocks), 0, 0)
    pf = struct.pack("<II4sIIIII", 32, 0x4, b"DX10", 0, 0, 0, 0, 0)
    caps = struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
    dx10 = struct.pack("<IIIII", 98, 3, 0, 1, 0)            # DXGI_FORMAT_BC7_UNORM, TEXTURE2D
    return hdr + pf + caps + dx10 + blocks


def pil_decode(blocks):
    """list of 16-byte blocks -> list of 16 x 4 uint8 arrays (texel = y*4 + x), decoded by Pillow."""
    from PIL import Image
    n = len(blocks)
    im = Image.open(io.BytesIO(dds_bc7(b"".join(blocks), n * 4, 4)))
    im.load()
    a = np.asarray(im.convert("RGBA"))
    return [a[:, 4 * i:4 * i + 4, :].reshape(16, 4).copy() for i in range(n)]


class _BW:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, val, bits):
        self.v |= (val & ((1 << bits) - 1)) << self.n
        self.n += bits

    def bytes(self):
        assert self.n == 128
        return self.v.to_bytes(16, "little")


def _mode3(part, ends, pbits, idxbits):          # 2 subsets: 0001 | 6-bit shape | 4 endpoints x 7 bits x RGB | 4 p-bits | 30 index bits
    w = _BW(); w.put(1 << 3, 4); w.put(part, 6)
    for ch in range(3):
        for e in range(4):
            w.put(ends[e][ch], 7)
    for e in range(4):
        w.put(pbits[e], 1)
    w.put(idxbits, 30)
    return w.bytes()


def _mode2(part, ends, idxbits):                 # 3 subsets: 001 | 6-bit shape | 6 endpoints x 5 bits x RGB | 29 index bits
    w = _BW(); w.put(1 << 2, 3); w.put(part, 6)
    for ch in range(3):
        for e in range(6):
            w.put(ends[e][ch], 5)
    w.put(idxbits, 29)
    return w.bytes()


def extract():
    K7, W7 = (0, 0, 0), (127, 127, 127)
    P2 = [[1 if t[0] > 128 else 0 for t in px] for px in pil_decode([_mode3(p, [K7, K7, W7, W7], [0, 0, 1, 1], 0) for p in range(64)])]
    A2 = []
    for p, px in enumerate(pil_decode([_mode3(p, [K7, W7, K7, W7], [0, 1, 0, 1], (1 << 30) - 1) for p in range(64)])):
        an = [i for i, t in enumerate(px) if t[0] < 200]                 # all index bits set: only the anchors (one bit short) are not at the far endpoint
        assert an[0] == 0 and len(an) == 2
        A2.append(an[1])
    K, R, Wh = (0, 0, 0), (31, 0, 0), (31, 31, 31)
    P3 = [[0 if t[0] < 128 else (1 if t[1] < 128 else 2) for t in px] for px in pil_decode([_mode2(p, [K, K, R, R, Wh, Wh], 0) for p in range(64)])]
    A3a, A3b = [], []
    for p, px in enumerate(pil_decode([_mode2(p, [K, Wh, K, Wh, K, Wh], (1 << 29) - 1) for p in range(64)])):
        an = [i for i, t in enumerate(px) if t[0] < 200]
        assert an[0] == 0 and len(an) == 3
        A3a.append([x for x in an[1:] if P3[p][x] == 1][0])
        A3b.append([x for x in an[1:] if P3[p][x] == 2][0])
    return dict(P2=[sum(1 << i for i, v in enumerate(r) if v) for r in P2], P3=[sum(v << (2 * i) for i, v in enumerate(r)) for r in P3],
                A2=A2, A3A=A3a, A3B=A3b)


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from unitygaussiansplatting_amd import bc7
    t = extract()
    for k, v in t.items():
        print(k, "=", [hex(x) for x in v] if k.startswith("P") else v)
        assert list(getattr(bc7, k)) == list(v), k
    print("bc7.py tables match Pillow")
