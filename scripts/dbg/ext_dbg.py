import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
os.environ["MSGL_EXTEND_NH"] = "2"; os.environ["MSGL_EXTEND_WIDE"] = "1"
import test_kernels_gpu as T
from oracle import torch_oracle as orc
ps = 1
prefix = [0, 0, ps * 2, 0, 64, 0]
inputs = [1, 31, 33, 128, 200, 300]
case = T.build_extend_case(16, 8, ps, prefix, inputs, seed=3 + 16 + ps)
out = T.run_extend(case, ps)
kc, vc = case["kv"][:, 0], case["kv"][:, 1]
ref = orc.paged_extend_ref(case["q"], kc, vc, case["r2t"], case["req_idx"], case["seq_lens"], case["prefix"], case["inp"], 128 ** -0.5)
err = (out.float() - ref.float()).abs().amax(dim=-1)  # [T, Hq]
start = 0
for b, n in enumerate(inputs):
    e = err[start:start + n]
    bad = (e > 2e-2).nonzero()
    print("req", b, "prefix", prefix[b], "n", n, "max", float(e.max()), "bad rows:", sorted(set(bad[:, 0].tolist()))[:40], "bad heads:", sorted(set(bad[:, 1].tolist())))
    start += n
