"""Split-size landscape of a ragged decode batch (mixed 1k-32k, unshared): GB/s per chunk size."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "none"]
import sweep_shapes as ss
from mini_sglang_amd.runtime import workload
bs = int(os.environ.get("BS", "256"))
lens = workload.mixed_lengths(bs, int(os.environ.get("LO", "1024")), int(os.environ.get("HI", "32768")), 3)
for chunk in (1024, 1536, 2048, 2560, 3072, 3584, 4096, 5120, 6144, 8192):
    n = sum(-(-l // chunk) for l in lens)
    print("chunk", chunk, "groups", n, end=" ")
    ss.run(f"ragged chunk {chunk}", 32, 8, 16, lens, layers=2, chunk=chunk, max_splits=64)

from mini_sglang_amd import ops
for cand in (1024, 1536, 2048, 2560, 3072, 3584, 4096, 5120, 6144, 8192):
    print("model", cand, ops.planned_makespan(lens, cand, 256, 128))
c = ops.pick_split_planned(lens, 64)
print("picked", c, end=" ")
ss.run(f"ragged picked {c}", 32, 8, 16, lens, layers=2, chunk=c, max_splits=64)
