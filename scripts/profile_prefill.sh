export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_prefill
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/e2e_prefill.py --steps 2 > $OUT/trace.log 2>&1 || exit 1
grep -v amdgpu.ids $OUT/trace.log | tail -2
python3 - <<'PY'
import csv, glob
f=glob.glob("/root/repo/gpurun_out/prof_prefill/trace/*/*kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'vectorized_elementwise' not in r["Name"] and 'distribution' not in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"])/3e6:9.2f} ms/prefill calls {int(r["Calls"])/3:7.1f} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:84]}')
PY
