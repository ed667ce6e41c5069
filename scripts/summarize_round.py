"""Markdown tables for DESIGN.md section 8.2 / profiles/README.md from a round's raw measurement files:

    python scripts/summarize_round.py profiles/r05_ab_rounds.log profiles/r05_bench_c2.json [profiles/r05_bench_c3.json ...]
The A/B log holds one JSON line per (configuration, build / sort mode, repetition) -- scripts/ab_rounds.sh; the bench files one bench.py line each."""
import json, os, sys
from collections import defaultdict


def lines(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")]


ab = defaultdict(lambda: defaultdict(list))
for d in lines(sys.argv[1]):
    who = ("previous round (" + d["lib"] + ")") if d.get("lib", "default") != "default" else ("HEAD, " + str(d.get("sort_mode", d.get("sort", "full"))) + " sort")
    ab[d["cfg"]][who].append(d)
print("| Config | build / mode | ms/frame (median region, per repetition) | keys | depth sort | view | bin | pair sort | blend | P | tile |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for cfg, byw in ab.items():
    base = None
    for who, ds in byw.items():
        w = [d["wall_med"] for d in ds]
        m = sum(w) / len(w)
        if base is None:
            base = m
        st = lambda k: (round(sum(d[k] for d in ds) / len(ds) * 1e3, 1) if all(isinstance(d.get(k), (int, float)) for d in ds) else
                        (round(sum(d["onesweep_depth"] for d in ds) / len(ds) * 1e3, 1) if k == "sort" else "-"))
        print(f"| {cfg} | {who} | {' / '.join(f'{x:.4f}' for x in w)} ({100 * (m / base - 1):+.1f} %) | {st('calc_distances')} | {st('sort')} | {st('calc_view')} | {st('bin')} | "
              f"{st('pair_sort')} | {st('blend')} | {ds[0]['P'] / 1e6:.2f} M | {ds[0]['tile']} |")
print()
print("| Config | headline mode | ms/frame | Msplats/s | full-sort mode ms | P | visible | keys | depth sort | view | bin | pair sort | blend | resolve | cross-check | oracle parity |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for path in sys.argv[2:]:
    d = lines(path)[-1]
    s = d["stages"]
    g = lambda k: round(s[k]["ms"] * 1e3, 1)
    full = d.get("modes", {}).get("full", {}).get("ms_per_step")
    cc = d.get("sort_mode_cross_check")
    par = d.get("parity_vs_oracle")
    print(f"| {os.path.basename(path)} | {d['config']['sort_mode']} | {d['ms_per_step']} | {d['value']} | {full} | {d['config']['tile_pairs_P'] / 1e6:.2f} M | "
          f"{d['config']['visible_splats'] / 1e6:.2f} M | {g('calc_distances')} | {g('sort')} | {g('calc_view')} | {g('bin')} | {g('pair_sort')} | {g('blend')} | {g('resolve')} | "
          f"{'ok' if cc and cc.get('ok') else cc} | {'ok' if par and par.get('within_bar') and par.get('order_bit_exact') and par.get('visible_mode', {}).get('order_is_visible_subsequence_of_oracle_order', True) else par} |")
    for k in ("roofline", "roofline_blend", "roofline_streaming"):
        r = d.get(k) or {}
        print(f"<!-- {k}: {r.get('kernel')} bound={r.get('bound')} achieved={r.get('achieved')} {r.get('unit')} peak={r.get('peak')} frac={r.get('frac')} avg_launch_ms={r.get('avg_launch_ms')} traffic={r.get('traffic')} -->")
