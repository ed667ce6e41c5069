"""Turns the rocprofv3 outputs of scripts/profile_round.sh (gpurun_out/prof/) into the files kept under profiles/:
  profiles/<tag>_rocprof_kernel_stats_<cfg>.txt   per-kernel time summary (--kernel-trace --stats)
  profiles/hbm_traffic.json                       HBM bytes per launch per kernel from the FETCH_SIZE / WRITE_SIZE PMC passes,
                                                  corrected with factors calibrated on known-byte-count kernels in the same session
Usage: python scripts/pmc_traffic.py <tag> [config] [visible|full] [key suffix] [steps] [warmup]
(the sort mode the bench ran in; stored under "<config>_visible" / "<config>" + the suffix, e.g. "C2_visible@s20w5" for the frames of the driver's
`bench.py --steps 20 --warmup 5`; bench.py picks the entry whose pair count matches its frame).  If the run directory also holds an SQ_INSTS_VALU pass,
profiles/valu_insts.json gets the matching entry."""
import csv, glob, json, os, re, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
config = sys.argv[2] if len(sys.argv) > 2 else "C2"
mode = sys.argv[3] if len(sys.argv) > 3 else "visible"
ksuf = sys.argv[4] if len(sys.argv) > 4 else ""
steps = sys.argv[5] if len(sys.argv) > 5 else "50"
warm = sys.argv[6] if len(sys.argv) > 6 else "10"
key = config + ("_visible" if mode == "visible" else "") + ksuf
P = os.path.join(ROOT, "gpurun_out", "prof_" + key)
if not os.path.isdir(P): P = os.path.join(ROOT, "gpurun_out", "prof")
# the frame the counters were collected on (bench.py refuses to pair a stored traffic with a frame whose pair count differs by > 2 %)
bench_P = bench_V = None
try:
    bl = [l for l in open(os.path.join(P, "bench_FETCH_SIZE.json")) if l.startswith("{")]
    bj = json.loads(bl[-1]); bench_P = bj["config"]["tile_pairs_P"]; bench_V = bj["config"]["visible_splats"]
except Exception:
    pass

def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name).replace("gs::", "")

def counters(d):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(P, d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out

# ---- calibration: every calib_* kernel moves exactly 1 GiB (scripts/probes/pmc_calib.hip); counters are in KiB
GIB_KIB = float(1 << 20)
cf, cw = counters("calib_FETCH_SIZE"), counters("calib_WRITE_SIZE")
cal = {}
for k in ("calib_read_b32", "calib_read_b128"):
    if k in cf: cal[k] = {"FETCH_SIZE_KiB": sum(cf[k]) / len(cf[k]), "true_over_counter": GIB_KIB / (sum(cf[k]) / len(cf[k]))}
# random gathers (round 5): what FETCH_SIZE tallies per LONE request -- 2,200,000 8-byte / 6,131,954 4-byte gathers (scripts/probes/pmc_calib.hip)
for k, cnt in (("calib_gather_b64", 2200000), ("calib_gather_b32", 6131954)):
    if k in cf: cal[k] = {"FETCH_SIZE_KiB": sum(cf[k]) / len(cf[k]), "bytes_tallied_per_gather": sum(cf[k]) / len(cf[k]) * 1024 / cnt,
                          "note": "a coalesced stream is tallied at half its bytes (x2.0); a lone gather at ~42 B, i.e. at face value (factor 1.0 .. 64/42 = 1.53 if every request moved a whole 64-byte sector)"}
for k in ("calib_write_b32", "calib_write_b128", "calib_scatter_runs32"):
    if k in cw: cal[k] = {"WRITE_SIZE_KiB": sum(cw[k]) / len(cw[k]), "true_over_counter": GIB_KIB / (sum(cw[k]) / len(cw[k]))}
    if k in cf: cal[k]["FETCH_SIZE_KiB_of_a_pure_write"] = sum(cf[k]) / len(cf[k])
# which calibration applies to which kernel's dominant access width
READ_CAL = {"onesweep_kernel": "calib_read_b32", "sort_keys_kernel": "calib_read_b32", "splat_depth_kernel": "calib_read_b32", "bin_emit_kernel": "calib_read_b32",
            "tile_ranges_kernel": "calib_read_b32", "calc_view_kernel": "calib_read_b128", "blend_kernel": "calib_read_b128", "resolve_kernel": "calib_read_b128",
            "visible_keys_kernel": "calib_read_b32", "tie_fix_kernel": "calib_read_b32", "vis_count_kernel": "calib_read_b128", "vis_offsets_kernel": "calib_read_b128",
            "vis_emit_kernel": "calib_read_b32"}
WRITE_CAL = {"onesweep_kernel": "calib_scatter_runs32", "sort_keys_kernel": "calib_write_b32", "bin_emit_kernel": "calib_write_b32",
             "calc_view_kernel": "calib_write_b128", "blend_kernel": "calib_write_b128", "resolve_kernel": "calib_write_b128",
             "visible_keys_kernel": "calib_write_b32", "vis_count_kernel": "calib_write_b128", "vis_offsets_kernel": "calib_write_b128", "vis_emit_kernel": "calib_write_b128"}
# kernels whose reads are dominated by LONE random gathers: the streaming factor (x2.0) over-counts them (profiles/r05_calib_gather.txt); their
# traffic is reported as a range -- fetch counter at face value (x1.0) .. at the streaming factor (x2.0) -- and `hbm_bytes_per_launch` is the LOWER end
GATHER_KERNELS = {"bin_emit_kernel", "vis_count_kernel"}
def by_base(c):
    out = defaultdict(list)
    for k, v in c.items():
        out[re.sub(r"<.*", "", k)] += v
    return out
bf, bw = by_base(counters("pmc_FETCH_SIZE")), by_base(counters("pmc_WRITE_SIZE"))
kernels = {}
for k in sorted(set(bf) | set(bw)):
    base = k
    if not (base.endswith("_kernel")): continue
    rf = cal.get(READ_CAL.get(base, "calib_read_b128"), {}).get("true_over_counter", 2.0)
    wf = cal.get(WRITE_CAL.get(base, "calib_write_b128"), {}).get("true_over_counter", 1.0)
    f_kib = sum(bf.get(k, [0])) / max(1, len(bf.get(k, [0])))
    w_kib = sum(bw.get(k, [0])) / max(1, len(bw.get(k, [0])))
    kernels[base] = {"launches_sampled": len(bf.get(k, [])), "FETCH_SIZE_KiB_per_launch": round(f_kib, 1), "WRITE_SIZE_KiB_per_launch": round(w_kib, 1),
                     "read_factor": round(rf, 3), "write_factor": round(wf, 3),
                     "hbm_bytes_per_launch": int((f_kib * rf + w_kib * wf) * 1024)}
    if base in GATHER_KERNELS:
        kernels[base].update({"read_factor": 1.0, "hbm_bytes_per_launch": int((f_kib * 1.0 + w_kib * wf) * 1024), "hbm_bytes_per_launch_upper": int((f_kib * rf + w_kib * wf) * 1024),
                              "note": "gather-dominated reads: FETCH_SIZE at face value (lower bound) .. at the streaming factor (upper bound); profiles/r05_calib_gather.txt"})
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
entry = {"config": config, "sort_mode": mode, "tile_pairs_P": bench_P, "visible_splats": bench_V,
         "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --config {config} --sort-mode {mode} --steps {steps} --warmup {warm}`, {tag}",
         "units": "counters are KiB; true bytes = counter * factor * 1024, factors calibrated on 1-GiB kernels of the same access width in the same session",
         "calibration": cal, "kernels": kernels}
allcfg = {}
if os.path.exists(tpath):
    try:
        old = json.load(open(tpath))
        allcfg = old.get("configs", {}) if "configs" in old else ({old["config"]: old} if "config" in old else {})
    except Exception:
        allcfg = {}
allcfg[key] = entry
json.dump({"configs": allcfg}, open(tpath, "w"), indent=1)

# ---- kernel time summary
rows = []
for f in glob.glob(os.path.join(P, "stats", "*", "*_kernel_stats.csv")):
    rows += list(csv.DictReader(open(f)))
trace = defaultdict(lambda: [0, 0, 0, 0])
for f in glob.glob(os.path.join(P, "stats", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        t = trace[short(r["Kernel_Name"])]
        t[0] = max(t[0], int(r.get("VGPR_Count", 0) or 0)); t[1] = max(t[1], int(r.get("SGPR_Count", 0) or 0)); t[2] = max(t[2], int(r.get("LDS_Block_Size", 0) or 0))
        t[3] = max(t[3], int(r.get("Grid_Size", 0) or r.get("Grid_Size_X", 0) or 0))
outp = os.path.join(ROOT, "profiles", f"{tag}_rocprof_kernel_stats_{key.lower().replace('@', '_')}.txt")
with open(outp, "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --config {config} --sort-mode {mode} --steps {steps} --warmup {warm} --cpu-baseline off --repeats 1   ({tag}; P = {bench_P}, visible = {bench_V})\n")
    o.write(f"{'kernel':34s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'grid':>9s}\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        n = short(r["Name"]); t = trace.get(n, [0, 0, 0, 0])
        o.write(f"{n[:34]:34s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3:11.1f} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f} {t[0]:5d} {t[1]:5d} {t[2]:6d} {t[3]:9d}\n")
    o.write("\n# HBM traffic per launch from the PMC passes (profiles/hbm_traffic.json)\n")
    for k, v in kernels.items():
        o.write(f"{k:34s} fetch {v['FETCH_SIZE_KiB_per_launch']:>11.1f} KiB x{v['read_factor']:<5}  write {v['WRITE_SIZE_KiB_per_launch']:>11.1f} KiB x{v['write_factor']:<5} -> {v['hbm_bytes_per_launch']/1e6:9.1f} MB\n")
    o.write("\n# the same per template instantiation (onesweep: <8, true> = the depth sort's gather pass, <8, false> = its other passes, <6|7, false> = pair sort)\n")
    ff, fw = counters("pmc_FETCH_SIZE"), counters("pmc_WRITE_SIZE")
    for k in sorted(set(ff) | set(fw)):
        if "<" not in k: continue
        base = re.sub(r"<.*", "", k)
        rf = cal.get(READ_CAL.get(base, "calib_read_b128"), {}).get("true_over_counter", 2.0)
        wf = cal.get(WRITE_CAL.get(base, "calib_write_b128"), {}).get("true_over_counter", 1.0)
        f_kib = sum(ff.get(k, [0])) / max(1, len(ff.get(k, [0]))); w_kib = sum(fw.get(k, [0])) / max(1, len(fw.get(k, [0])))
        o.write(f"{k[:34]:34s} fetch {f_kib:>11.1f} KiB  write {w_kib:>11.1f} KiB -> {(f_kib * rf + w_kib * wf) * 1024 / 1e6:9.1f} MB  ({len(ff.get(k, []))} launches)\n")
    o.write("\n# calibration (1 GiB each)\n")
    for k, v in cal.items(): o.write(f"{k:26s} {v}\n")
print(open(outp).read())

# ---- SQ_INSTS_VALU of the VALU-bound kernels on the same frames (profile_round.sh's last pass) -> profiles/valu_insts.json
sq = counters("pmc_SQ_INSTS_VALU")
ks = {}
for k, v in by_base(sq).items():
    if k in ("blend_kernel", "calc_view_kernel") and v:
        ks[k] = {"valu_wave_insts": int(sum(v) / len(v)), "launches_sampled": len(v)}
if ks:
    vpath = os.path.join(ROOT, "profiles", "valu_insts.json")
    allv = {}
    if os.path.exists(vpath):
        try: allv = json.load(open(vpath)).get("configs", {})
        except Exception: allv = {}
    allv[config + ksuf] = {"config": config, "tile_pairs_P": bench_P, "visible_splats": bench_V, "kernels": ks,
                           "source": f"rocprofv3 --pmc SQ_INSTS_VALU on `bench.py --config {config} --sort-mode {mode} --steps {steps} --warmup {warm}` (scripts/profile_round.sh), {tag}: mean wave-level VALU instructions per launch"}
    json.dump({"configs": allv}, open(vpath, "w"), indent=1)
    print("valu_insts.json:", config + ksuf, ks)
