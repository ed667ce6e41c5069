"""Kernel durations with frames in flight, from a rocprofv3 --kernel-trace csv of `bench.py --sort-mode visible_in_flight --repeats 1`:
per kernel the median duration one frame at a time (the `visible` mode's frames: one queue) and in flight (two queues), the share of the
in-flight wall time during which 1 / 2 kernels run, and the busy time per queue.
    python scripts/inflight_trace.py <kernel_trace.csv> [fraction of the one-at-a-time part to drop, default 0.4]
(lanes inside the library: trace `--sort-mode all` -- the owner's queue runs the full, reference-shaped and visible modes one after the other before the lanes start -- and
pass 0.85: the tail of that part is the visible mode's frames)"""
import collections, csv, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gs::", "")
    return re.sub(r"[<(].*", "", n)
for r in rows:
    r["s"], r["e"], r["k"], r["q"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0")
rows.sort(key=lambda r: r["s"])
# the in-flight part = the tail of the trace from the first kernel on a second queue that runs calc_view
by_q = collections.defaultdict(list)
for r in rows:
    if r["k"].startswith("calc_view"):
        by_q[r["q"]].append(r["s"])
qs = sorted(by_q, key=lambda q: by_q[q][0])
if len(qs) < 2:
    sys.exit("only one queue runs calc_view: not an in-flight trace")
t_split = by_q[qs[1]][0]
seq = [r for r in rows if r["e"] < t_split and r["q"] == qs[0]]
fl = [r for r in rows if r["s"] >= t_split]
# drop warm-up: keep the last 60 % of each part
seq = seq[int(len(seq) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.4)):]; fl = fl[int(len(fl) * 0.4):]
def med(part):
    d = collections.defaultdict(list)
    for r in part: d[r["k"]].append((r["e"] - r["s"]) / 1e3)
    return {k: (statistics.median(v), len(v)) for k, v in d.items()}
ms, mf = med(seq), med(fl)
print(f"{'kernel':28s} {'alone us':>9s} {'in flight us':>12s} {'x':>6s}")
for k in sorted(mf, key=lambda k: -mf[k][0] * mf[k][1]):
    if k in ms: print(f"{k[:28]:28s} {ms[k][0]:9.1f} {mf[k][0]:12.1f} {mf[k][0] / ms[k][0]:6.2f}")
# concurrency profile of the in-flight part
ev = sorted([(r["s"], 1) for r in fl] + [(r["e"], -1) for r in fl])
t0, t1 = ev[0][0], ev[-1][0]
lvl, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
tot = t1 - t0
print("kernels running at once (share of the in-flight wall time): " + ", ".join(f"{k}: {v / tot:.3f}" for k, v in sorted(hist.items())))
nframes = sum(1 for r in fl if r["k"].startswith("blend"))
print(f"in-flight part: {nframes} frames in {tot / 1e6:.3f} ms = {tot / 1e6 / max(nframes, 1):.4f} ms per frame; sum of kernel durations per frame {sum((r['e'] - r['s']) for r in fl) / 1e6 / max(nframes, 1):.4f} ms")
