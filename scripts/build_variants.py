"""Builds A/B variants of the library in parallel:  python scripts/build_variants.py name:DEF1,DEF2=3 name2:... """
import os, sys, subprocess
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import build as B
def one(spec):
    name, _, defs = spec.partition(":")
    out_dir = os.path.join(B.HERE, "variants"); os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + ".so")
    cmd = [B.hipcc()] + B.FLAGS + ["-w"] + ["-D" + d for d in defs.split(",") if d] + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-ldl", "-lz", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr[-2000:] if r.returncode else ""
with ThreadPoolExecutor(8) as ex:
    for name, rc, err in ex.map(one, sys.argv[1:]):
        print(name, "ok" if rc == 0 else "FAILED\n" + err, flush=True)
