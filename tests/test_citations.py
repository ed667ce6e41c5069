"""Every `File.ext:line[-line]` citation of the reference in the header, the design documents, the kernels, the host layer, the oracle and the tests names a
file that exists under /root/reference and lines that file has.  (CPU suite, this container only: the GPU box has no reference tree -- skipped there.)"""
import collections
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CITE = re.compile(r"\b([A-Za-z0-9_]+\.(?:cs|hlsl|compute|shader|cginc))[:](\d+)(?:[-–](\d+))?")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "package")), reason="no reference tree on this box")
def test_reference_citations_point_at_existing_lines():
    index = collections.defaultdict(list)
    for dp, _, fs in os.walk(REF):
        for f in fs:
            index[f].append(os.path.join(dp, f))
    lengths = {}
    files = [os.path.join(ROOT, f) for f in ("include/gsplat_c.h", "DESIGN.md", "INTEGRATION.md", "README.md", "bench.py", "__graft_entry__.py")]
    for pat in ("unitygaussiansplatting_amd/csrc/*", "unitygaussiansplatting_amd/*.py", "unitygaussiansplatting_amd/dotnet/*.cs", "oracle/*.cpp", "oracle/*.h",
                "oracle/ref_build/*.py", "oracle/ref_build/*.h", "oracle/ref_build/*.cpp", "tests/*.py", "tests/*.c"):
        files += glob.glob(os.path.join(ROOT, pat))
    seen, bad = 0, []
    for fn in files:
        if not os.path.isfile(fn):
            continue
        for m in CITE.finditer(open(fn, errors="replace").read()):
            name, a, b = m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))
            seen += 1
            if name not in index:
                bad.append((os.path.relpath(fn, ROOT), m.group(0), "no such file in the reference")); continue
            if name not in lengths:
                lengths[name] = max(sum(1 for _ in open(p, errors="replace")) for p in index[name])
            if not (1 <= a <= b <= lengths[name]):
                bad.append((os.path.relpath(fn, ROOT), m.group(0), f"the file has {lengths[name]} lines"))
    assert seen > 200, seen
    assert not bad, bad
