"""Gaps between consecutive kernels of the frame loop, from a rocprofv3 --kernel-trace csv (no hipEvents in the traced run).
    python scripts/trace_gaps.py <kernel_trace.csv> [frames_to_use]"""
import collections, csv, re, statistics, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
use = int(sys.argv[2]) if len(sys.argv) > 2 else 40

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("gs::", "")
    return re.sub(r"\(.*", "", n)

# the frame loop = the tail of the trace; a frame starts with sort_keys_kernel
starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("sort_keys_kernel")]
starts = starts[-(use + 1):]
gaps = collections.defaultdict(list)
frame_ms, busy_ms = [], []
for a, b in zip(starts[:-1], starts[1:]):
    fr = rows[a:b + 1]                       # includes the first kernel of the next frame
    frame_ms.append((int(fr[-1]["Start_Timestamp"]) - int(fr[0]["Start_Timestamp"])) / 1e6)
    busy_ms.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fr[:-1]) / 1e6)
    for x, y in zip(fr[:-1], fr[1:]):
        gaps[(short(x["Kernel_Name"]), short(y["Kernel_Name"]))].append((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3)
print(f"frames {len(frame_ms)}: start-to-start median {statistics.median(frame_ms):.4f} ms, sum of kernel durations median {statistics.median(busy_ms):.4f} ms")
tot = 0.0
for k, v in gaps.items():
    if len(v) >= len(frame_ms) // 2:
        m = statistics.median(v); tot += m * (len(v) / len(frame_ms))
        print(f"{m:8.2f} us x{len(v) / len(frame_ms):.0f}  {k[0][:32]:32s} -> {k[1][:32]}")
print(f"sum of median gaps per frame {tot:.1f} us")
