"""The `layers` object of bench.py on its own: HBM bandwidth of the fused layer kernels at prefill (16384 tokens) and
decode (256 tokens) sizes, Qwen3-8B shapes, median of 7 graph replays per op."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

torch.cuda.set_device(0)
args = bench.parse()
res = bench.layer_kernels(args, 1, 0)
print(res.get("error", ""))
for name, v in res["ops"].items():
    print(f"{name:28s} {v['us_per_launch']['median']:9.2f} us  [{v['us_per_launch']['min']:.2f}, {v['us_per_launch']['max']:.2f}]  "
          f"{v['GBps']:7.1f} GB/s  {v['frac_of_hbm_peak']:.3f}" + (f"  cpu oracle {json.dumps(v['cpu_oracle'])}" if "cpu_oracle" in v else ""), flush=True)
