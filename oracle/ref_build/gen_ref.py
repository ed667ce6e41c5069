#!/usr/bin/env python3
"""gen_ref.py <reference root> <out dir> -- make the reference's shader TEXT compilable as C++ (oracle/_ref, test infrastructure).

Reads, at build time, from <reference root>/package/Shaders/ (never copied into the repository; <out dir> is a scratch
directory that the Makefile deletes after compiling):
    GaussianSplatting.hlsl                 whole file
    SplatUtilities.compute                 lines 37-252: the two #includes, the uniforms, FloatToSortableUint, CSSetIndices,
                                           CSCalcDistances, DecomposeCovariance, IsSplatCut, CSCalcViewData
                                           (line 36 includes DeviceRadixSort.hlsl -- wave intrinsics + groupshared, not
                                           compiled here: the sort's contract is "stable, ascending" and is tested as such)
    RenderGaussianSplats.shader            the CGPROGRAM ... ENDCG block (vert + frag)
    GaussianComposite.shader               the CGPROGRAM ... ENDCG block (vert + frag)

Every statement stays the reference's.  The only rewrites are syntactic, each one a regular expression below:
    R1  [numthreads(...)]                          removed (a D3D attribute; the harness loops over the thread ids)
    R2  ": SV_xxx" / ": COLORn" / ": TEXCOORDn"    removed (HLSL semantics)
    R3  "out T x" / "inout T x" parameters         -> "T& x"
    R4  "(StructType)0"                            -> hlsl_zero<StructType>()   (C++ has no scalar -> struct cast)
    R5  unsuffixed floating literals               -> suffixed with f (an HLSL literal is a float, a C++ one a double:
                                                      without this C++ would evaluate `cov._m00 += 0.3` in double)
The asserts pin the line ranges to the reference revision the citations in oracle/gs_oracle.cpp were made against.
"""
import os
import re
import sys

RULES = [
    ("R1", re.compile(r"\[numthreads\([^\]]*\)\]"), ""),
    ("R2", re.compile(r"\s*:\s*(?:SV_\w+|COLOR\d*|TEXCOORD\d*)\b"), ""),
    ("R3", re.compile(r"(?<=[(,])(\s*)(?:in)?out\s+(\w+)\s+(\w+)"), r"\1\2& \3"),
    ("R4", re.compile(r"\((SplatData|SplatViewData|v2f)\)\s*0\b"), r"hlsl_zero<\1>()"),
    ("R5", re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])"), r"\1f"),
]


def rewrite(text, counts):
    for name, rx, rep in RULES:
        text, n = rx.subn(rep, text)
        counts[name] = counts.get(name, 0) + n
    return text


def cgprogram(text, path):
    m = re.search(r"^CGPROGRAM\s*$(.*?)^ENDCG\s*$", text, re.S | re.M)
    assert m, f"no CGPROGRAM block in {path}"
    return m.group(1)


def main():
    ref, out = sys.argv[1], sys.argv[2]
    sh = os.path.join(ref, "package", "Shaders")
    os.makedirs(out, exist_ok=True)
    counts = {}

    gs = open(os.path.join(sh, "GaussianSplatting.hlsl")).read()
    assert "float3 CalcCovariance2D(" in gs and "SplatData LoadSplatData(uint idx)" in gs
    open(os.path.join(out, "GaussianSplatting.hlsl"), "w").write(rewrite(gs, counts))

    cs = open(os.path.join(sh, "SplatUtilities.compute")).read().split("\n")
    assert cs[35].strip() == '#include "DeviceRadixSort.hlsl"', cs[35]
    assert cs[36].strip() == '#include "GaussianSplatting.hlsl"', cs[36]
    assert cs[39].strip() == "float4x4 _MatrixObjectToWorld;", cs[39]
    assert cs[251].strip() == "}" and cs[254].strip() == "RWByteAddressBuffer _DstBuffer;", (cs[251], cs[254])
    open(os.path.join(out, "SplatUtilities_37_252.inc"), "w").write(rewrite("\n".join(cs[36:252]) + "\n", counts))

    for name in ("RenderGaussianSplats", "GaussianComposite"):
        path = os.path.join(sh, name + ".shader")
        open(os.path.join(out, name + ".inc"), "w").write(rewrite(cgprogram(open(path).read(), path), counts))
    print("gen_ref: rewrites applied", counts)


if __name__ == "__main__":
    main()
