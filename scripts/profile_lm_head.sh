#!/bin/bash
# rocprofv3 passes on the fused greedy head (run from the repo root on the GPU box): kernel trace of
# scripts/bench_lm_head.py at batch 32 and 64 on the 151 936 x 4096 head, then FETCH_SIZE in a pass of its own.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LH_QUICK=1
OUT=gpurun_out/prof_lm_head
SUM=gpurun_out/profile_lm_head.txt
rm -rf $OUT; mkdir -p $OUT; : > $SUM
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_lm_head.py > $OUT/trace.log 2>&1 || { echo "trace failed" | tee -a $SUM; exit 1; }
grep '^{' $OUT/trace.log | tee -a $SUM
python3 - $OUT/trace <<'PY' | tee -a $SUM
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "lm_head" in n or "argmax" in n or "Cijk" in n or "gemm" in n.lower():
            print(f"  {n[:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.2f} us min {float(r['MinNs'])/1e3:10.2f} max {float(r['MaxNs'])/1e3:10.2f}")
PY
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scripts/bench_lm_head.py > $OUT/fetch.log 2>&1 || { echo "fetch pass failed" | tee -a $SUM; exit 2; }
python3 - $OUT/fetch <<'PY' | tee -a $SUM
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lm_head_argmax_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    m = sum(v) / len(v)
    print(f"  {k}: FETCH_SIZE mean {m:.0f} KiB over {len(v)} dispatches -> 2 x = {2 * m * 1024 / 1e9:.3f} GB per launch (weights: 1.245 GB)")
PY
