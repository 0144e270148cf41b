"""Pins the CPU oracle (oracle/torch_oracle.py and the oracle-backed harness) to outputs of the reference's own
torch-native path, bit for bit: the fixtures in tests/golden/ were produced by oracle/gen_golden.py from the
reference's RMSNorm / SiluAndMul / RotaryEmbedding modules, its TorchNativeAttnBackend and its
ModelRunner + BatchInfo generation loop.  Runs on the CPU (no GPU, no HIP calls)."""
import pytest
import torch

import golden_cases as gc
from oracle import gen_golden as gg
from oracle import modules_oracle as mo
from oracle import torch_oracle as orc


def test_ops_fixture_matches_oracle():
    fx = gc.load("ops.pt")
    for rows, cols, seed in gg.OPS_CASES["rmsnorm"]:
        x, w = gc.rmsnorm_inputs(rows, cols, seed)
        assert torch.equal(orc.rmsnorm_ref(x, w, 1e-6), fx[f"rmsnorm/{rows}x{cols}/{seed}"])
    for rows, cols, seed in gg.OPS_CASES["fused_add_rmsnorm"]:
        x, r, w = gc.fused_inputs(rows, cols, seed)
        y, r2 = orc.rmsnorm_ref(x, w, 1e-6, residual=r)
        assert torch.equal(y, fx[f"fused_add_rmsnorm/{rows}x{cols}/{seed}/y"])
        assert torch.equal(r2, fx[f"fused_add_rmsnorm/{rows}x{cols}/{seed}/residual"])
    for rows, n, seed in gg.OPS_CASES["silu_and_mul"]:
        assert torch.equal(orc.silu_and_mul_ref(gc.silu_inputs(rows, n, seed)), fx[f"silu_and_mul/{rows}x{n}/{seed}"])
    for T, Hq, Hkv, D, seed in gg.OPS_CASES["rope"]:
        pos, q, k = gc.rope_inputs(T, Hq, Hkv, D, seed)
        cache = orc.build_cos_sin_cache(D, 40960, 1e6)
        key = f"rope/{T}x{Hq}x{Hkv}x{D}/{seed}"
        assert torch.equal(cache[[0, 1, 40959]], fx[key + "/cache_rows"])
        qo, ko = orc.rope_ref(pos, q, k, cache, D)
        assert torch.equal(qo, fx[key + "/q"]) and torch.equal(ko, fx[key + "/k"])


def test_attention_fixture_matches_oracle():
    fx = gc.load("attention.pt")
    for ci, case in enumerate(gg.ATTN_DECODE_CASES):
        c = gc.decode_inputs(case)
        orc.store_kv_ref(c["kc"], c["vc"], c["loc"], c["k_new"], c["v_new"])
        out = orc.paged_decode_ref(c["q"].view(-1, c["Hq"], 128), c["kc"], c["vc"], c["pt"].req_to_token, c["req_idx"],
                                   c["pt"].seq_lens, 128 ** -0.5)
        assert torch.equal(out.reshape(len(case[3]), -1), fx[f"decode/{ci}"]), f"decode case {ci}"
    for ci, case in enumerate(gg.ATTN_EXTEND_CASES):
        c = gc.extend_inputs(case)
        orc.store_kv_ref(c["kc"], c["vc"], c["loc"], c["k_new"], c["v_new"])
        out = orc.paged_extend_ref(c["q"].view(-1, c["Hq"], 128), c["kc"], c["vc"], c["pt"].req_to_token, c["req_idx"],
                                   c["pt"].seq_lens, c["prefix"], c["inp"], 128 ** -0.5)
        got = out.reshape(out.shape[0], -1)[:, ::gg.EXTEND_COL_STRIDE]
        assert torch.equal(got, fx[f"extend/{ci}"]), f"extend case {ci}"


@pytest.mark.parametrize("name", ["tiny_ps1", "tiny_ps16"])
def test_e2e_fixture_matches_oracle_harness(name):
    """The same stack as the product (BatchInfo page-table writes, Qwen3-shaped modules, runner) with oracle
    layers and oracle attention reproduces the reference's generation exactly: tokens, logits, slot
    assignments and the final page table."""
    fx = gc.load(f"e2e_{name}.pt")
    got = gc.run_e2e(name, "cpu", layers=mo, backend_factory=mo.OracleAttnBackend)
    for key in ("tokens", "margins", "top1_logits", "logits_slices", "logits_wide", "out_cache_loc_first", "out_cache_loc_decode",
                "req_to_token_rows"):
        assert torch.equal(got[key], fx[key]), key
    assert got["req_to_page_matches_tokens"]


def test_e2e_8b_layer_shapes_fixture_matches_oracle_harness():
    """Qwen3-8B widths on four layers (page_size 16, batch 3): the first six steps of the oracle stack on the host equal the
    reference's run bit for bit (the whole 40 steps take the host several minutes; the GPU test covers all of them)."""
    name, steps = "qwen3_8b_l4_ps16", 6
    fx = gc.load(f"e2e_{name}.pt")
    got = gc.run_e2e(name, "cpu", layers=mo, backend_factory=mo.OracleAttnBackend, new_tokens=steps)
    assert torch.equal(got["tokens"], fx["tokens"][:, :steps])
    for key in ("margins", "top1_logits", "logits_slices", "logits_wide"):
        assert torch.equal(got[key], fx[key][:steps]), key
    assert torch.equal(got["out_cache_loc_first"], fx["out_cache_loc_first"])
    assert torch.equal(got["out_cache_loc_decode"], fx["out_cache_loc_decode"][: steps - 1])


@pytest.mark.parametrize("name", ["tiny_ps1", "tiny_ps16", "qwen3_0.6b_ps1", "qwen3_8b_l4_ps16"])
def test_prepare_decode_oracle_matches_reference_slots(name):
    """The decode-step preparation restated in the oracle reproduces the slots the REFERENCE assigned in its own
    run of the case (fixture fields out_cache_loc_first / out_cache_loc_decode / req_to_token_rows)."""
    fx = gc.load(f"e2e_{name}.pt")
    _, ps, (_, plens), _, _, _, _, _ = gg.E2E_CASES[name]
    first = fx["out_cache_loc_first"].tolist()
    pages = 1024 // ps
    used = len(first) if ps == 1 else sum(-(-n // ps) for n in plens)   # pages the extend step took, in order
    free_pages = list(range(1, pages + 1))
    head = used
    r2t = torch.zeros((len(plens), 2048), dtype=torch.int32)
    seq, last, pos = list(plens), [], 0
    for i, n in enumerate(plens):
        r2t[i, :n] = torch.tensor(first[pos: pos + n], dtype=torch.int32)
        pos += n
        last.append(first[pos - 1])
    rows = list(range(len(plens)))
    for step, want in enumerate(fx["out_cache_loc_decode"].tolist()):
        positions, loc, head = orc.prepare_decode_ref(rows, seq, last, free_pages, head, r2t, ps)
        assert loc == want, f"step {step}"
        assert positions == [s - 1 for s in seq]
    got_rows = torch.cat([r2t[i, : seq[i]] for i in rows])
    assert torch.equal(got_rows, fx["req_to_token_rows"].to(torch.int32))
