"""BC7 (BPTC) block codec for the VeryLow presets colour texture (GaussianSplatAsset.cs:56,169: ColorFormat.BC7 =
GraphicsFormat.RGBA_BC7_UNorm; written by EditorUtility.CompressTexture, GaussianSplatAssetCreator.cs:893-910).

The format is public (Khronos Data Format Specification, "BPTC"); its three fixed tables -- the 64 two-subset and 64
three-subset partition patterns and the anchor ("fix-up") indices -- are reproduced below.  They were not typed from
memory: tests/bc7_tables_from_pil.py extracts them from an INDEPENDENT decoder that happens to be installed here (Pillows
BcnDecode, by decoding probe blocks), and tests/test_bc7.py checks this decoder against Pillow on random blocks of every
mode, so the tables and the bit layout are pinned by a second implementation.

decode_block / decode_texture are the plain-Python / numpy reference used by the tests and the importers PSNR check; the
shipped decoders are gs_device_math.h (HIP) and oracle/gs_oracle.cpp (checker).  encode_texture_mode6 is the importers
encoder: every block in mode 6 (one subset, 7.7.7.7 endpoints + p-bit, 4-bit indices) -- a valid BC7 stream, not Unitys
(EditorUtility.CompressTextures output cannot be reproduced; any conforming decoder reads both).
"""
from __future__ import annotations

import numpy as np

# partition of texel i (row-major in the 4x4 block) for the 64 two-subset shapes: bit i of P2[shape] = subset
P2 = [0xcccc, 0x8888, 0xeeee, 0xecc8, 0xc880, 0xfeec, 0xfec8, 0xec80, 0xc800, 0xffec, 0xfe80, 0xe800, 0xffe8, 0xff00, 0xfff0, 0xf000, 0xf710, 0x8e, 0x7100, 0x8ce, 0x8c, 0x7310, 0x3100, 0x8cce, 0x88c, 0x3110, 0x6666, 0x366c, 0x17e8, 0xff0, 0x718e, 0x399c, 0xaaaa, 0xf0f0, 0x5a5a, 0x33cc, 0x3c3c, 0x55aa, 0x9696, 0xa55a, 0x73ce, 0x13c8, 0x324c, 0x3bdc, 0x6996, 0xc33c, 0x9966, 0x660, 0x272, 0x4e4, 0x4e40, 0x2720, 0xc936, 0x936c, 0x39c6, 0x639c, 0x9336, 0x9cc6, 0x817e, 0xe718, 0xccf0, 0xfcc, 0x7744, 0xee22]
# three-subset shapes: 2 bits per texel, texel i at bits 2i..2i+1
P3 = [0xaa685050, 0x6a5a5040, 0x5a5a4200, 0x5450a0a8, 0xa5a50000, 0xa0a05050, 0x5555a0a0, 0x5a5a5050, 0xaa550000, 0xaa555500, 0xaaaa5500, 0x90909090, 0x94949494, 0xa4a4a4a4, 0xa9a59450, 0x2a0a4250, 0xa5945040, 0xa425054, 0xa5a5a500, 0x55a0a0a0, 0xa8a85454, 0x6a6a4040, 0xa4a45000, 0x1a1a0500, 0x50a4a4, 0xaaa59090, 0x14696914, 0x69691400, 0xa08585a0, 0xaa821414, 0x50a4a450, 0x6a5a0200, 0xa9a58000, 0x5090a0a8, 0xa8a09050, 0x24242424, 0xaa5500, 0x24924924, 0x24499224, 0x50a50a50, 0x500aa550, 0xaaaa4444, 0x66660000, 0xa5a0a5a0, 0x50a050a0, 0x69286928, 0x44aaaa44, 0x66666600, 0xaa444444, 0x54a854a8, 0x95809580, 0x96969600, 0xa85454a8, 0x80959580, 0xaa141414, 0x96960000, 0xaaaa1414, 0xa05050a0, 0xa0a5a5a0, 0x96000000, 0x40804080, 0xa9a8a9a8, 0xaaaaaa44, 0x2a4a5254]
# anchor texel of subset 1 (two subsets) / subsets 1 and 2 (three subsets); subset 0s anchor is texel 0
A2 = [15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 2, 8, 2, 2, 8, 8, 15, 2, 8, 2, 2, 8, 8, 2, 2, 15, 15, 6, 8, 2, 8, 15, 15, 2, 8, 2, 2, 2, 15, 15, 6, 6, 2, 6, 8, 15, 15, 2, 2, 15, 15, 15, 15, 15, 2, 2, 15]
A3A = [3, 3, 15, 15, 8, 3, 15, 15, 8, 8, 6, 6, 6, 5, 3, 3, 3, 3, 8, 15, 3, 3, 6, 10, 5, 8, 8, 6, 8, 5, 15, 15, 8, 15, 3, 5, 6, 10, 8, 15, 15, 3, 15, 5, 15, 15, 15, 15, 3, 15, 5, 5, 5, 8, 5, 10, 5, 10, 8, 13, 15, 12, 3, 3]
A3B = [15, 8, 8, 3, 15, 15, 3, 8, 15, 15, 15, 15, 15, 15, 15, 8, 15, 8, 15, 3, 15, 8, 15, 8, 3, 15, 6, 10, 15, 15, 10, 8, 15, 3, 15, 10, 10, 8, 9, 10, 6, 15, 8, 15, 3, 6, 6, 8, 15, 3, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 3, 15, 15, 8]
W2 = (0, 21, 43, 64)
W3 = (0, 9, 18, 27, 37, 46, 55, 64)
W4 = (0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64)
# mode: (subsets, partition bits, rotation bits, index-selection bits, colour bits, alpha bits, per-endpoint p-bits, shared p-bits, index bits, secondary index bits)
MODES = ((3, 4, 0, 0, 4, 0, 1, 0, 3, 0), (2, 6, 0, 0, 6, 0, 0, 1, 3, 0), (3, 6, 0, 0, 5, 0, 0, 0, 2, 0), (2, 6, 0, 0, 7, 0, 1, 0, 2, 0),
         (1, 0, 2, 1, 5, 6, 0, 0, 2, 3), (1, 0, 2, 0, 7, 8, 0, 0, 2, 2), (1, 0, 0, 0, 7, 7, 1, 0, 4, 0), (2, 6, 0, 0, 5, 5, 1, 0, 2, 0))


def _weights(bits):
    return W2 if bits == 2 else (W3 if bits == 3 else W4)


def subset_of(ns: int, shape: int, texel: int) -> int:
    if ns == 1:
        return 0
    if ns == 2:
        return (P2[shape] >> texel) & 1
    return (P3[shape] >> (2 * texel)) & 3


def anchors_of(ns: int, shape: int):
    if ns == 1:
        return (0,)
    if ns == 2:
        return (0, A2[shape])
    return (0, A3A[shape], A3B[shape])


def decode_block(block: bytes) -> np.ndarray:
    """16 bytes -> 16 x 4 uint8 (texel = y * 4 + x, channels RGBA)."""
    v = int.from_bytes(block, "little")
    pos = 0

    def get(n):
        nonlocal pos
        r = (v >> pos) & ((1 << n) - 1)
        pos += n
        return r
    mode = 0
    while mode < 8 and not (v >> mode) & 1:
        mode += 1
    out = np.zeros((16, 4), np.uint8)
    if mode == 8:
        return out                                   # reserved: all zero
    pos = mode + 1
    ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2 = MODES[mode]
    shape = get(pb)
    rot = get(rb)
    isel = get(isb)
    ne = 2 * ns
    ep = [[0, 0, 0, 255] for _ in range(ne)]
    for ch in range(3):
        for e in range(ne):
            ep[e][ch] = get(cb)
    if ab:
        for e in range(ne):
            ep[e][3] = get(ab)
    if epb:
        for e in range(ne):
            p = get(1)
            for ch in range(4 if ab else 3):
                ep[e][ch] = (ep[e][ch] << 1) | p
    elif spb:
        for s in range(ns):
            p = get(1)
            for e in (2 * s, 2 * s + 1):
                for ch in range(3):
                    ep[e][ch] = (ep[e][ch] << 1) | p
    cbits = cb + (1 if (epb or spb) else 0)
    abits = ab + (1 if (epb and ab) else 0)
    for e in range(ne):
        for ch in range(3):
            x = ep[e][ch] << (8 - cbits)
            ep[e][ch] = x | (x >> cbits)
        if ab:
            x = ep[e][3] << (8 - abits)
            ep[e][3] = x | (x >> abits)
    anchors = anchors_of(ns, shape)
    idx1 = []
    for t in range(16):
        s = subset_of(ns, shape, t)
        idx1.append(get(ib - 1 if t == anchors[s] else ib))
    idx2 = None
    if ib2:
        idx2 = [get(ib2 - 1 if t == 0 else ib2) for t in range(16)]
    for t in range(16):
        s = subset_of(ns, shape, t)
        e0, e1 = ep[2 * s], ep[2 * s + 1]
        ci, cbt = idx1[t], ib
        ai, abt = (idx2[t], ib2) if ib2 else (idx1[t], ib)
        if isel:
            ci, cbt, ai, abt = ai, abt, ci, cbt
        wc, wa = _weights(cbt)[ci], _weights(abt)[ai]
        px = [((64 - wc) * e0[c] + wc * e1[c] + 32) >> 6 for c in range(3)] + [((64 - wa) * e0[3] + wa * e1[3] + 32) >> 6]
        if rot:
            px[3], px[rot - 1] = px[rot - 1], px[3]
        out[t] = px
    return out


def decode_texture(data: np.ndarray, width: int, height: int) -> np.ndarray:
    """BC7 blocks (row-major, width/4 per row) -> height x width x 4 uint8."""
    d = np.ascontiguousarray(data, np.uint8).reshape(-1, 16)
    bw = width // 4
    out = np.zeros((height, width, 4), np.uint8)
    for b in range(len(d)):
        by, bx = divmod(b, bw)
        out[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4] = decode_block(d[b].tobytes()).reshape(4, 4, 4)
    return out


def encode_texture_mode6(rgba: np.ndarray) -> np.ndarray:
    """height x width x 4 float in [0,1] -> BC7 blocks, every block in mode 6 (vectorised).  Endpoints = the block's
    per-channel min / max quantised to 7 bits + a p-bit chosen per endpoint to minimise its own error; 4-bit indices
    by projection on the endpoint axis; the anchor rule (index of texel 0 < 8) is met by swapping the endpoints."""
    H, W, _ = rgba.shape
    assert H % 4 == 0 and W % 4 == 0
    x = np.clip(np.asarray(rgba, np.float32), 0.0, 1.0) * np.float32(255.0)
    blk = x.reshape(H // 4, 4, W // 4, 4, 4).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 4)        # [B, 16, 4]
    lo, hi = blk.min(1), blk.max(1)

    def quant(v):                                            # 8-bit value -> best (7-bit, p-bit) with a shared p per endpoint
        best, berr = None, None
        for p in (0, 1):
            q = np.clip(np.floor((v - p) / 2.0 + 0.5), 0, 127)
            rec = q * 2 + p
            err = ((rec - v) ** 2).sum(1)
            if best is None:
                best, berr, bp = q, err, np.full(len(v), p)
            else:
                m = err < berr
                best[m], berr[m], bp[m] = q[m], err[m], p
        return best.astype(np.int64), bp.astype(np.int64)
    q0, p0 = quant(lo)
    q1, p1 = quant(hi)
    e0 = (q0 * 2 + p0[:, None]).astype(np.float32)
    e1 = (q1 * 2 + p1[:, None]).astype(np.float32)
    axis = e1 - e0
    den = (axis * axis).sum(1)
    t = ((blk - e0[:, None, :]) * axis[:, None, :]).sum(2) / np.maximum(den, 1e-6)[:, None]
    w = np.asarray(W4, np.float32) / 64.0
    idx = np.abs(t[:, :, None] - w[None, None, :]).argmin(2).astype(np.int64)                    # [B, 16]
    swap = idx[:, 0] >= 8                                    # anchor: texel 0's index must have its top bit clear
    idx[swap] = 15 - idx[swap]
    q0s, q1s, p0s, p1s = np.where(swap[:, None], q1, q0), np.where(swap[:, None], q0, q1), np.where(swap, p1, p0), np.where(swap, p0, p1)
    out = np.zeros((len(blk), 16), np.uint8)
    for b in range(len(blk)):                                # bit packing per block (offline importer path)
        v, pos = 1 << 6, 7
        for ch in range(4):
            v |= int(q0s[b, ch]) << pos; pos += 7
            v |= int(q1s[b, ch]) << pos; pos += 7
        v |= int(p0s[b]) << pos; pos += 1
        v |= int(p1s[b]) << pos; pos += 1
        v |= int(idx[b, 0]) << pos; pos += 3
        for k in range(1, 16):
            v |= int(idx[b, k]) << pos; pos += 4
        out[b] = np.frombuffer(v.to_bytes(16, "little"), np.uint8)
    return out.reshape(-1)
