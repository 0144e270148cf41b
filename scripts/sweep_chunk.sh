for c in 256 512 1024 2048 4096 8192; do
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --chunk $c --max-splits 32 > gpurun_out/sw_$c.log 2>&1; python - <<PY
import json
l=open("gpurun_out/sw_$c.log").read().strip().splitlines()[-1]
try:
    d=json.loads(l); print("chunk",$c,"value",d["value"],"ms",d["ms_per_step"])
except Exception as e: print("chunk",$c,"ERR",l[-300:])
PY
done
