#!/bin/bash
# lane utilisation of the VALU-bound kernels: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) -- is there lane waste a compaction could recover?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lanes; rm -rf $O; mkdir -p $O; export PYTHONPATH=$R
rocprofv3 -L 2>/dev/null | grep -o "SQ_THREAD_CYCLES_VALU\|SQ_ACTIVE_INST_VALU\|SQ_INSTS_VALU_[A-Z0-9_]*\|SQ_VALU_[A-Z_0-9]*" | sort -u | head -40
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/p -- python $R/scripts/bench_stages.py C2 20 > $O/log 2>&1
python - <<'PY'
import csv, glob, collections, re
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/root/repo/gpurun_out/pmc_lanes/p/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"[<(].*","",r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("gs::",""))
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    if m.get("SQ_ACTIVE_INST_VALU"):
        print(f"{k[:28]:28s} insts {m.get('SQ_INSTS_VALU',0)/1e6:8.2f} M  thread_cycles/(active_inst*64) = {m.get('SQ_THREAD_CYCLES_VALU',0)/(m['SQ_ACTIVE_INST_VALU']*64):.3f}   thread_cycles/(insts*64*4)= {m.get('SQ_THREAD_CYCLES_VALU',0)/(max(m.get('SQ_INSTS_VALU',1),1)*256):.3f}")
PY
