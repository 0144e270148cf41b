export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_e2e_small
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 scripts/e2e_try.py ${E2E_ARGS:---model 0.6b --bs 64 --ctx 2048 --ps 1} --steps 6 > $OUT/trace.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f=glob.glob("/root/repo/gpurun_out/prof_e2e_small/trace/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if 'argmax' in r["Kernel_Name"]]
a,b=idx[-4],idx[-1]
seg=rows[a+1:b+1]; steps=3
acc=collections.defaultdict(lambda:[0,0])
gap=0
for x,y in zip(seg[:-1],seg[1:]): gap+=max(0,int(y["Start_Timestamp"])-int(x["End_Timestamp"]))
for r in seg:
    d=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); n=r["Kernel_Name"][:80]
    acc[n][0]+=d; acc[n][1]+=1
span=(int(seg[-1]["End_Timestamp"])-int(seg[0]["Start_Timestamp"]))/steps/1e6
tot=sum(v[0] for v in acc.values())/steps/1e6
print(f"step span {span:.3f} ms, kernel sum {tot:.3f} ms, gaps {gap/steps/1e6:.3f} ms, kernels/step {len(seg)/steps:.0f}")
for n,(d,c) in sorted(acc.items(), key=lambda x:-x[1][0])[:16]:
    print(f"{d/steps/1e6:8.3f} ms/step {c/steps:6.1f} calls avg {d/c/1e3:7.1f} us  {n}")
PY
