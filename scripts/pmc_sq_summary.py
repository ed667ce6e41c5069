"""Per-kernel means of the SQ counters collected by scripts/pmc_sq.sh (gpurun_out/pmc_sq/p*/)."""
import csv, glob, os, re, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "pmc_sq")
def short(n): return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", "")).replace("gs::", "")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(P, "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
for k, cs in sorted(acc.items()):
    if not k.endswith("_kernel") and "<" not in k: continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(f"== {k}  (n={len(next(iter(cs.values())))})")
    wc = m.get("SQ_WAVE_CYCLES", 0)
    for c in names:
        if c in m:
            extra = f"  ({100 * m[c] / wc:5.1f} % of wave cycles)" if wc and c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS") else ""
            print(f"   {c:24s} {m[c]:16.0f}{extra}")

# ---- profiles/valu_insts.json: SQ_INSTS_VALU per launch of the VALU-bound kernels, for bench.py's VALU roofline (python scripts/pmc_sq_summary.py C2 r05)
if len(sys.argv) > 1:
    import json
    cfg = sys.argv[1]; tag = sys.argv[2] if len(sys.argv) > 2 else ""
    P_ = V_ = None
    try:
        jl = [l for l in open(os.path.join(P, "p1.log")) if l.startswith("{")]
        jj = json.loads(jl[-1]); P_ = jj.get("P"); V_ = jj.get("V")
    except Exception:
        pass
    ks = {}
    for k, cs in acc.items():
        base = re.sub(r"<.*", "", k)
        if base in ("blend_kernel", "calc_view_kernel") and "SQ_INSTS_VALU" in cs:
            v = cs["SQ_INSTS_VALU"]
            ks[base] = {"valu_wave_insts": int(sum(v) / len(v)), "launches_sampled": len(v)}
    path = os.path.join(ROOT, "profiles", "valu_insts.json")
    allc = {}
    if os.path.exists(path):
        try: allc = json.load(open(path)).get("configs", {})
        except Exception: allc = {}
    allc[cfg] = {"config": cfg, "tile_pairs_P": P_, "visible_splats": V_, "kernels": ks,
                 "source": f"rocprofv3 --pmc SQ_INSTS_VALU ... on `scripts/bench_stages.py {cfg}` (scripts/pmc_sq.sh), {tag}: mean wave-level VALU instructions per launch"}
    json.dump({"configs": allc}, open(path, "w"), indent=1)
