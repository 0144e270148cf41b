# kernel-trace of the whole-model decode step (headline measurement skipped): where a step's time goes
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_e2e
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extend --e2e-steps 8 > $OUT/trace.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f=glob.glob("/root/repo/gpurun_out/prof_e2e/trace/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:25]:
    print(f'{float(r["TotalDurationNs"])/1e6:10.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% calls {r["Calls"]:>6} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:90]}')
PY
