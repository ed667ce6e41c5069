"""VGPR/SGPR/LDS/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage); the ISA is kept under /tmp/isa."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("/tmp/isa", exist_ok=True)
files = sys.argv[1:] or ["gs_sort", "gs_view", "gs_raster"]
for f in files:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fno-gpu-rdc",
           "-save-temps", "-c", "-Rpass-analysis=kernel-resource-usage", os.path.join(ROOT, "unitygaussiansplatting_amd", "csrc", f + ".hip"),
           "-o", f"/tmp/isa/{f}.o"] + [a for a in os.environ.get("EXTRA", "").split() if a]
    out = subprocess.run(cmd, cwd="/tmp/isa", capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r":\d+:\d+:\s*(?:remark:)?\s*([A-Za-z][A-Za-z \[\]/]*?):\s*(\S+)", line)
        if not m: continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]}
        else:
            cur[k] = v
        if k == "LDS Size [bytes/block]" or k.startswith("LDS Size"):
            print(f"{cur['name'][:40]:40s} vgpr {cur.get('VGPRs','?'):>4s} agpr {cur.get('AGPRs','?'):>3s} sgpr {cur.get('TotalSGPRs','?'):>4s} scratch {cur.get('ScratchSize [bytes/lane]','?'):>4s} occ {cur.get('Occupancy [waves/SIMD]','?'):>2s} lds {v}")
