# rocprofv3 evidence for this round's kernels; run from the repo root on the GPU box (through gpurun).  Every pass is a
# run of its own under a time limit; counters (--pmc) are never combined with a trace.  Summary: gpurun_out/profile_r03.txt
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
rm -rf $OUT; mkdir -p $OUT
SUM=gpurun_out/profile_r03.txt
: > $SUM
say() { echo "$@" | tee -a $SUM; }

stats() {  # kernel-trace summary lines of a run directory, kernels of this library only
  python3 - "$1" <<'PY' | tee -a gpurun_out/profile_r03.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "msgl::" in r["Name"]:
            print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>4} avg {float(r["AverageNs"])/1e3:10.2f} us min {float(r["MinNs"])/1e3:10.2f} max {float(r["MaxNs"])/1e3:10.2f}')
PY
}
counters() {  # mean counter values per dispatch of the kernels matching $2
  python3 - "$1" "$2" <<'PY' | tee -a gpurun_out/profile_r03.txt
import collections, csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    acc = collections.defaultdict(list); regs = None
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            regs = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
    print("  ", sys.argv[2], "(vgpr, agpr, sgpr, lds, scratch) =", regs, {k: round(sum(v[1:]) / max(1, len(v) - 1)) if len(v) > 1 else round(v[0]) for k, v in acc.items()})
PY
}

say "== 1. headline decode (Qwen3-8B, batch 256 x 8192, page_size 16): python3 bench.py, kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/head_trace -- python3 bench.py --no-cpu-baseline --no-e2e --no-configs --no-extend --no-layers --steps 40 > $OUT/head_trace.log 2>&1 || { say "trace failed"; exit 1; }
grep '^{' $OUT/head_trace.log | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('  bench line:', r['value'], r['unit'], r['roofline'])" | tee -a $SUM
stats $OUT/head_trace

say "== 2. headline decode: HBM traffic (FETCH_SIZE, WRITE_SIZE; tools/decode_bench, same kernel and shape)"
HEAD="--chunk 4096 --splits 2 --steps 4 --warmup 1"
mkdir -p gpurun_out/prof_decode; rm -rf gpurun_out/prof_decode/fetch gpurun_out/prof_decode/write
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_decode/fetch -- ./tools/decode_bench $HEAD > $OUT/fetch.log 2>&1 || { say "fetch pass failed"; exit 2; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_decode/write -- ./tools/decode_bench $HEAD > $OUT/write.log 2>&1 || { say "write pass failed"; exit 3; }
python3 scripts/pmc_traffic.py "${MSGL_COMMIT:-unknown}" | tee -a $SUM

say "== 3. decode with 8 query heads per kv head (matrix-core kernel)"
for cfg in "tp8_bs8_128k --hq 8 --hkv 1 --bs 8 --ctx 131072 --ps 64 --chunk 4096 --splits 32 --layers 4" "tp1_bs256_8k --hq 64 --hkv 8 --bs 256 --ctx 8192 --ps 16 --chunk 4096 --splits 2 --layers 2"; do
  set -- $cfg; name=$1; shift
  say "-- $name: $*"
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/g8_${name}_trace -- ./tools/decode_bench "$@" --steps 10 --warmup 2 > $OUT/g8_${name}_trace.log 2>&1 || say "trace failed"
  tail -1 $OUT/g8_${name}_trace.log | tee -a $SUM
  stats $OUT/g8_${name}_trace
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/g8_${name}_fetch -- ./tools/decode_bench "$@" --steps 3 --warmup 1 > $OUT/g8_${name}_fetch.log 2>&1 || say "fetch failed"
  counters $OUT/g8_${name}_fetch paged_decode_mfma
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/g8_${name}_sq -- ./tools/decode_bench "$@" --steps 3 --warmup 1 > $OUT/g8_${name}_sq.log 2>&1 || say "sq failed"
  counters $OUT/g8_${name}_sq paged_decode_mfma
done

say "== 4. extend (prefill) attention"
for cfg in "p16k --bs 1 --sq 16384 --prefix 0 --ps 16 --hq 32 --hkv 8" "tp8_4k_on_64k --bs 1 --sq 4096 --prefix 65536 --ps 64 --hq 8 --hkv 1"; do
  set -- $cfg; name=$1; shift
  say "-- $name: $*"
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ext_${name}_trace -- ./tools/extend_bench "$@" > $OUT/ext_${name}_trace.log 2>&1 || say "trace failed"
  tail -1 $OUT/ext_${name}_trace.log | tee -a $SUM
  stats $OUT/ext_${name}_trace
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/ext_${name}_p1 -- ./tools/extend_bench "$@" --steps 2 --warmup 1 > $OUT/ext_${name}_p1.log 2>&1 || say "p1 failed"
  counters $OUT/ext_${name}_p1 paged_extend
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/ext_${name}_p2 -- ./tools/extend_bench "$@" --steps 2 --warmup 1 > $OUT/ext_${name}_p2.log 2>&1 || say "p2 failed"
  counters $OUT/ext_${name}_p2 paged_extend
done
# matrix-core busy fraction of the 16k-prompt extend launch, as a record bench.py attaches to its extend figure
python3 - "$OUT/ext_p16k_p2" "${MSGL_COMMIT:-unknown}" <<'PY' | tee -a $SUM
import collections, csv, glob, json, sys
acc = collections.defaultdict(list); kernel = None
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "paged_extend" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); kernel = r["Kernel_Name"].split("(")[0]
if acc:
    mean = {k: sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0] for k, v in acc.items()}
    rec = {"shape": {"prompt": 16384, "prefix": 0, "page_size": 16, "hq": 32, "hkv": 8}, "kernel": kernel,
           "matrix_core_busy": round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * mean["SQ_BUSY_CU_CYCLES"]), 4),
           "mfma_instructions": round(mean["SQ_INSTS_MFMA"]), "lds_bank_conflict_cycles": round(mean["SQ_LDS_BANK_CONFLICT"]),
           "how": "rocprofv3 --pmc (one pass, no trace) on tools/extend_bench: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), mean per dispatch after the first",
           "commit": sys.argv[2]}
    json.dump(rec, open("gpurun_out/extend_counters.json", "w"), indent=1)
    print("  extend counters record:", json.dumps(rec))
PY
say "== 5. layer kernels (scripts/bench_layers.py: prefill 16384 tokens and decode 256 tokens, Qwen3-8B shapes): kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/layers_trace -- python3 scripts/bench_layers.py > $OUT/layers_trace.log 2>&1 || say "layers trace failed"
grep -v "^$" $OUT/layers_trace.log | grep "T16384\|T256" | tee -a $SUM
stats $OUT/layers_trace

say "== 6. decode at batch 1 x 131072 tokens (Qwen3-32B TP-8 shard, page_size 64): kernel trace by split size, then the in-kernel time line"
for cfg in "512 256" "1024 128" "256 512"; do
  set -- $cfg
  say "-- chunk $1 x $2 splits"
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bs1_c$1 -- ./tools/decode_bench --hq 8 --hkv 1 --bs 1 --ctx 131072 --ps 64 --chunk $1 --splits $2 --layers 8 --steps 40 --warmup 5 > $OUT/bs1_c$1.log 2>&1 || say "trace failed"
  tail -1 $OUT/bs1_c$1.log | tee -a $SUM
  stats $OUT/bs1_c$1
done
if [ -x tools/ko/decode_stamps ]; then scripts/stamp_decode.sh run 2>&1 | tee -a $SUM; fi

say "== 7. what issues beside a matrix instruction (tools/exp/mfma_coissue)"
if [ -x tools/exp/mfma_coissue ]; then timeout -k 5 60 tools/exp/mfma_coissue 2>&1 | tee -a $SUM; fi
if [ -x tools/ko/extend_clock ]; then
  say "== 8. extend steady-state loop: cycles per iteration by segment (stamped build), then undisturbed with the clock"
  timeout -k 10 120 tools/ko/extend_stamps --bs 2 --sq 16384 --steps 5 --warmup 2 2>&1 | tee -a $SUM
  timeout -k 10 120 tools/ko/extend_clock --bs 2 --sq 16384 --steps 40 --warmup 10 2>&1 | grep "whole loop\|TFLOPs" | tee -a $SUM
fi
say "done"
