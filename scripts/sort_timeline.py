"""Phase timeline of the Onesweep kernel's last launch (needs a library built with -DGS_EXP_SORT_TIMELINE; GSPLAT_LIB=...).
Sorts n random pairs once with 1 pass (key_bits=8) and prints, per phase, when partitions reach it (us since kernel start)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitygaussiansplatting_amd import _lib
from unitygaussiansplatting_amd.renderer import GpuContext, GpuSorting
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_131_954
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = GpuContext(0)
s = GpuSorting(ctx, n)
rng = np.random.default_rng(0)
k = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32); v = np.arange(n, dtype=np.uint32)
for _ in range(3):
    s.DispatchHost(k, v, bits)
lib = C.CDLL(_lib.LIB_PATH)
part = 512 * 16
parts = (n + part - 1) // part
buf = np.zeros((16384, 16), np.uint64)
assert lib.gs_debug_read_sort_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
t = buf[:parts].astype(np.int64)
t0 = t[:, 0].min()
us = (t[:, :10] - t0) / 100.0
names = ["start", "ticket", "keys in", "ranked", "scattered", "lookback(d0)", "lookback all", "keys out", "vals in LDS", "vals out"]
print(f"n={n} parts={parts}  (us since the first block started; 100 MHz clock)")
print(f"{'phase':14s} {'min':>8s} {'p10':>8s} {'median':>8s} {'p90':>8s} {'max':>8s}   {'dt median (from previous phase)':>s}")
for i, nm in enumerate(names):
    c = us[:, i]
    dt = np.median(us[:, i] - us[:, i - 1]) if i else 0.0
    print(f"{nm:14s} {c.min():8.2f} {np.percentile(c,10):8.2f} {np.median(c):8.2f} {np.percentile(c,90):8.2f} {c.max():8.2f}   {dt:8.2f}")
order = np.argsort(us[:, 0])
print("start time by partition index (every 64th):", np.round(us[::64, 0], 1).tolist())
print("end time by partition index (every 64th):  ", np.round(us[::64, 9], 1).tolist())
r = t[:, 10:13]
print("look-back of digit 0: rounds  median %d p90 %d max %d | words consumed median %d p90 %d max %d | empty polls median %d p90 %d max %d" % (
    np.median(r[:, 0]), np.percentile(r[:, 0], 90), r[:, 0].max(), np.median(r[:, 1]), np.percentile(r[:, 1], 90), r[:, 1].max(),
    np.median(r[:, 2]), np.percentile(r[:, 2], 90), r[:, 2].max()))
print("per partition (every 32nd): idx, scattered, lookback done, rounds, words, empty polls")
for i in range(0, parts, 32):
    print(f"  {i:5d} {us[i,4]:7.2f} {us[i,5]:7.2f} {r[i,0]:5d} {r[i,1]:5d} {r[i,2]:6d}")
