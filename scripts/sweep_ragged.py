"""Split-size landscape of a ragged decode batch (mixed 1k-32k, unshared): GB/s per chunk size."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "none"]
import sweep_shapes as ss
from mini_sglang_amd.runtime import workload
lens = workload.mixed_lengths(256, 1024, 32768, 3)
for chunk in (512, 640, 768, 896, 1024, 1152, 1280, 1536, 1792, 2048, 2560, 3072, 4096):
    n = sum(-(-l // chunk) for l in lens)
    print("chunk", chunk, "groups", n, end=" ")
    ss.run(f"ragged chunk {chunk}", 32, 8, 16, lens, layers=2, chunk=chunk, max_splits=64)
