# Anatomy of the batch-1 x 131072-token decode launch (Qwen3-32B TP-8 shard: 8 q / 1 kv head, page_size 64): rocprofv3 kernel
# trace of the standalone driver (durations of the main kernel and of the split merge) for a few split sizes.
# Run from the repo root on the GPU box; output: gpurun_out/decode_bs1.txt
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_bs1
rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/decode_bs1.txt
: > $SUM
for cfg in "512 256" "1024 128" "256 512" "2048 64"; do
  set -- $cfg
  echo "== chunk $1 x $2 splits" | tee -a $SUM
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c$1 -- ./tools/decode_bench --hq 8 --hkv 1 --bs ${BS:-1} --ctx 131072 --ps 64 --chunk $1 --splits $2 --layers 8 --steps 40 --warmup 5 > $OUT/c$1.log 2>&1 || { echo "trace failed" | tee -a $SUM; exit 1; }
  tail -1 $OUT/c$1.log | tee -a $SUM
  python3 - $OUT/c$1 <<'PY' | tee -a $SUM
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "msgl::" in r["Name"]:
            print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>4} avg {float(r["AverageNs"])/1e3:8.2f} us min {float(r["MinNs"])/1e3:8.2f} max {float(r["MaxNs"])/1e3:8.2f}')
PY
done
