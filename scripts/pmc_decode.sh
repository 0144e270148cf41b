export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
for cfg in "g4 --hq 32 --hkv 8 --bs 256 --ctx 8192 --chunk 4096 --splits 2" "g8 --hq 64 --hkv 8 --bs 64 --ctx 8192 --chunk 1024 --splits 8" "g8tp8 --hq 8 --hkv 1 --bs 8 --ctx 131072 --ps 64 --chunk 2048 --splits 64"; do
  set -- $cfg; name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/$name -- ./tools/decode_bench --steps 3 --warmup 1 --layers 2 "$@" > $OUT/$name.log 2>&1 || echo "fail $name"
  tail -1 $OUT/$name.log
done
python3 - <<'PY'
import csv, glob, collections
for name in ("g4","g8","g8tp8"):
    fs=glob.glob(f"/root/repo/gpurun_out/pmc/{name}/*/*counter_collection.csv")
    if not fs: print(name,"no csv"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "paged_decode_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(name, {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
