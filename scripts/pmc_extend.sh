# SQ counters of the extend kernel (standalone driver), two passes of <= 8 SQ counters each.
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_extend
rm -rf $OUT; mkdir -p $OUT
ARGS="${EXT_ARGS:---bs 2 --sq 8192}"
timeout -k 10 120 ./tools/extend_bench $ARGS > $OUT/driver.log 2>&1 || exit 4
cat $OUT/driver.log
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- ./tools/extend_bench $ARGS > $OUT/trace.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/p1 -- ./tools/extend_bench $ARGS --steps 2 --warmup 1 > $OUT/p1.log 2>&1 || echo "fail p1"
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/p2 -- ./tools/extend_bench $ARGS --steps 2 --warmup 1 > $OUT/p2.log 2>&1 || echo "fail p2"
tail -3 $OUT/p2.log
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("/root/repo/gpurun_out/pmc_extend/trace/*/*kernel_stats.csv"):
    for l in open(f):
        if "extend" in l: print(l.strip())
for name in ("p1","p2"):
    fs=glob.glob(f"/root/repo/gpurun_out/pmc_extend/{name}/*/*counter_collection.csv")
    if not fs: print(name,"no csv"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "paged_extend" in r["Kernel_Name"] and "merge" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            regs=(r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
    print(name, regs, {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
