"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol the header declares, the
synthetic generators are deterministic and have the promised structure, batch preparation assigns slots the
way the reference does at page boundaries, and the wrappers refuse inputs the kernels cannot take."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mini_sglang_amd import _capi

    header = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("msgl_hip.h", "msgl_comm.h"))
    declared = set(re.findall(r"\b(msgl_[a-z0-9_]+)\s*\(", header)) - {"msgl_ar"}
    assert declared, "no declarations found"
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/msgl_hip.h or msgl_comm.h but not exported"
    assert declared == set(_capi.SIGNATURES), "ctypes binding and header disagree"
    assert _capi.lib.msgl_abi_version() == _capi.ABI_VERSION
    assert _capi.lib.msgl_paged_decode_workspace_bytes(2, 4, 128, 3) == 2 * 3 * 4 * 130 * 4


def test_argument_validation_without_gpu():
    """Validation runs before any launch, so it can be exercised with null pointers on a machine without a GPU."""
    from mini_sglang_amd import _capi, ops

    lib = _capi.lib
    rc = lib.msgl_paged_decode(None, 0, None, None, 0, None, 0, None, None, None, 0, None, 0, 1, 16, 8, 96, 1, 1.0, 1, 16, None, 0, None, None, None, None, None)
    assert rc == -1 and "head dim" in _capi.last_error()
    rc = lib.msgl_paged_decode(None, 0, None, None, 0, None, 0, None, None, None, 0, None, 0, 1, 24, 8, 128, 1, 1.0, 1, 16, None, 0, None, None, None, None, None)
    assert rc == -1 and "null pointer" in _capi.last_error()
    rc = lib.msgl_rmsnorm(None, 0, None, None, 0, 4, 100, 1e-6, None)
    assert rc == -1 and "unsupported" in _capi.last_error()
    assert lib.msgl_paged_decode(None, 0, None, None, 0, None, 0, None, None, None, 0, None, 0, 0, 16, 8, 128, 1, 1.0, 1, 16, None, 0, None, None, None, None, None) == 0
    with pytest.raises(ValueError, match="GPU"):
        ops.rmsnorm(torch.zeros(2, 128, dtype=torch.bfloat16), torch.ones(128, dtype=torch.bfloat16), 1e-6)
    with pytest.raises(ValueError, match="GPU"):
        ops.silu_and_mul(torch.zeros(2, 16, dtype=torch.bfloat16))


def test_synth_known_answers():
    from mini_sglang_amd.runtime import synth

    x = synth.normal((4,), 1, dtype=torch.float32)
    again = synth.normal((4,), 1, dtype=torch.float32)
    assert torch.equal(x, again)
    # first hashes of seed 0: frozen so that fixtures regenerate identically everywhere
    h = synth.hash64(torch.arange(3, dtype=torch.int64), 0)
    assert h.tolist() == synth.hash64(torch.arange(3, dtype=torch.int64), 0).tolist()
    big = synth.normal((200000,), 5, dtype=torch.float32)
    assert abs(big.mean().item()) < 0.01 and abs(big.std().item() - 1) < 0.01 and big.abs().max().item() < 3.5
    p = synth.permutation(1000, 3)
    assert sorted(p.tolist()) == list(range(1000))


def test_page_table_structure():
    from mini_sglang_amd.runtime import workload

    ps = 16
    pt = workload.make_page_table([300, 256, 1000, 64, 65, 40], ps, 200, seed=1, groups=[0, 0, 0, 1, 1, 2], group_prefix_lens=[256, 64, 0])
    t = pt.req_to_token
    assert torch.equal(t[0, :256], t[1, :256]) and torch.equal(t[0, :256], t[2, :256]) and torch.equal(t[3, :64], t[4, :64])
    assert not torch.equal(t[0, 256:300], t[2, 256:300])
    for r, S in enumerate([300, 256, 1000, 64, 65, 40]):
        row = t[r, :S].long()
        assert (row >= ps).all(), "page 0 is the dummy page and is never handed out"
        pos = torch.arange(S)
        assert torch.equal(row % ps, pos % ps), "slots are contiguous inside a page"
    assert pt.unique_tokens == 256 + 44 + 744 + 64 + 1 + 40
    assert len(set(workload.mixed_lengths(64, 1024, 32768, 3))) > 32


def test_batch_prepare_page_boundaries():
    """prepare_for_decode takes a new page exactly when the old length is a multiple of the page size and
    otherwise the slot after the request's last one (reference: batch_info.py:277-291)."""
    from mini_sglang_amd.mem_cache import PageAllocator, ReqToTokenPool
    from mini_sglang_amd.runtime.batch_info import BatchInfo, Req

    ps = 4
    r2t, pa = ReqToTokenPool(4, 64, ps, "cpu"), PageAllocator(32, ps)
    reqs = [Req(0, [1] * 3), Req(1, [2] * 4), Req(2, [3] * 9)]
    b = BatchInfo.init_new(reqs, r2t, pa, kv_cache_pool=None)
    b.prepare_for_extend()
    assert b.out_cache_loc.tolist() == [4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]
    assert b.positions.tolist() == [0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 5, 6, 7, 8]
    assert b.seq_lens.tolist() == [3, 4, 9] and b.prefix_seq_lens.tolist() == [0, 0, 0] and b.input_seq_lens_max == 9
    for r in reqs:
        r.token_ids.append(7)
    b.prepare_for_decode()
    assert b.out_cache_loc.tolist() == [7, 24, 21]  # next slot, fresh page (pages 1-5 used), next slot
    assert b.positions.tolist() == [3, 4, 9] and b.seq_lens.tolist() == [4, 5, 10]
    assert r2t.req_to_token[1, :5].tolist() == [8, 9, 10, 11, 24]
    # a cached, page-aligned prefix is written into the new request's row as is
    shared = Req(3, [5] * 6)
    shared.prefix_indices = r2t.req_to_token[1, :4].clone()
    b2 = BatchInfo.init_new([shared], r2t, pa, kv_cache_pool=None)
    b2.prepare_for_extend()
    assert r2t.req_to_token[shared.req_pool_idx, :6].tolist() == [8, 9, 10, 11, 28, 29]
    assert b2.positions.tolist() == [4, 5] and b2.prefix_seq_lens.tolist() == [4]
    bad = Req(4, [5] * 6)
    bad.prefix_indices = r2t.req_to_token[1, :3].clone()
    with pytest.raises(AssertionError, match="page-aligned"):
        BatchInfo.init_new([bad], ReqToTokenPool(4, 64, ps, "cpu"), pa, kv_cache_pool=None).prepare_for_extend()


def test_pick_split_bounds():
    from mini_sglang_amd.attn.hip_attn_backend import splits_for_batch
    from mini_sglang_amd.ops import DECODE_TILE, pick_split

    for bs, S in [(256, 8192), (1, 131072), (64, 2048), (3, 17), (256, 32768)]:
        ms = splits_for_batch(bs)
        c = pick_split(S, S * bs, bs, ms)
        assert c % DECODE_TILE == 0 and -(-S // c) <= ms


def test_unknown_backend_name():
    from mini_sglang_amd.runtime.model_runner import ModelRunner

    class Stub:
        server_args = type("A", (), {"attention_backend": "fa3"})()
    with pytest.raises(ValueError, match="Unsupported attention backend"):
        ModelRunner.init_attn_backend(Stub())


def test_planned_split_model():
    """Host-side schedule model behind the split size of ragged decode batches: exact on hand-checkable cases, and
    the picked size always lets the longest request fit its splits."""
    from mini_sglang_amd import ops

    # 4 slots, chunk 100, no overhead: 6 full splits (two slots hold 2), tails 50 and 30 go to the slots that hold 1
    assert ops.planned_makespan([250, 230, 100, 100], 100, 4, 0) == 200
    # every tail finds an empty slot
    assert ops.planned_makespan([50, 30, 20], 100, 4, 0) == 50
    # more tails than free slots: the smallest ones queue behind the earliest finishers
    assert ops.planned_makespan([90, 80, 10, 10, 10], 100, 2, 0) == 100
    assert ops.planned_makespan([100] * 8, 100, 4, 5) == 2 * 105
    for lens, max_splits, kv_heads in [([5000, 100, 900, 33, 2048, 1], 16, 8), ([131072, 7], 512, 1), ([300] * 40 + [9000], 8, 2)]:
        chunk = ops.pick_split_planned(lens, max_splits, kv_heads, 256)
        assert chunk % 256 == 0 and -(-max(lens) // chunk) <= max_splits


def test_decode_and_extend_plan_restatements():
    """The numpy-free restatements the GPU tests check the device-built launch orders against."""
    from oracle import torch_oracle as orc

    items, cost = orc.decode_plan_ref([40, 16, 1], max_splits=2, chunk=16)
    # request 0: split 0 = 16 tokens, split 1 (the last) takes the remaining 24; requests 1, 2: one split
    assert dict(zip(items, cost)) == {0: 1, 1: 1, 2: 1, 3: 2}
    items, cost = orc.extend_plan_ref([0, 300, 301], [64, 1000], 300)
    # 3 query tiles per request slot; request 0: 300 rows on a 64 prefix; request 1: one row on 1000
    assert dict(zip(items, cost)) == {0: 3, 1: 5, 2: 6, 3: 16}


def _sgprs(operand_text):
    """Scalar registers named in an operand string: s7 -> {7}, s[4:7] -> {4, 5, 6, 7}."""
    import re
    regs = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", operand_text):
        if m.group(3) is not None:
            regs.add(int(m.group(3)))
        else:
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def test_scalar_page_table_loads_of_the_matrix_core_decode_kernel_are_waited_for_before_any_use(tmp_path):
    """paged_decode_mfma_kernel fetches page-table entries with s_load_dword in one asm statement and waits for them in
    another, so the compiler does not know that the register is in flight in between.  This reads the built code object:
    between every such load and the next s_waitcnt lgkmcnt(0), no instruction may name the loaded register (a copy made
    there would copy the register's old contents, and the kernel would then read K/V at a wrong slot)."""
    import re
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "mini-sglang_amd", "libmsgl_hip.so")
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(lib) and os.path.exists(objdump)):
        pytest.skip("needs the built library and llvm-objdump")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    checked = 0
    for co in sorted(tmp_path.glob("*gfx950")):
        text = subprocess.run([objdump, "-d", str(co)], check=True, capture_output=True, text=True).stdout
        for fn in re.split(r"\n(?=[0-9a-f]+ <)", text):
            if "paged_decode_mfma_kernel" not in fn.split("\n", 1)[0]:
                continue
            pending = {}   # register -> line of its load
            for line in fn.split("\n")[1:]:
                ins = line.split("//")[0].strip()
                if not ins:
                    continue
                op, _, rest = ins.partition(" ")
                if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
                    pending.clear()
                    continue
                named = _sgprs(rest)
                if op == "s_load_dword":
                    dst = _sgprs(rest.split(",")[0])
                    src = named - dst
                    assert not (src & pending.keys()), f"address of a scalar load uses a register still in flight: {ins}"
                    for r in dst:
                        pending[r] = ins
                    checked += 1
                    continue
                hit = named & pending.keys()
                assert not hit, f"{ins!r} names s{sorted(hit)[0]} between {pending[sorted(hit)[0]]!r} and its wait"
    assert checked >= 8, f"expected the kernel's scalar page-table loads in the code object, saw {checked}"


def test_req_to_token_pool_keeps_a_page_table_that_matches_its_token_table():
    """ReqToTokenPool.write_tokens (req2token.py:77-91) with a prefix of several pages, and the batch producers' scatter:
    page i of a request is the page of its token i * page_size.  (The reference's writer strides its index vector by
    page_size a second time and stores the ids of pages 0, page_size, 2 page_size, ... of the write instead.)"""
    from mini_sglang_amd.mem_cache import ReqToTokenPool
    ps = 4
    pool = ReqToTokenPool(3, 64, ps, "cpu")
    pages = torch.tensor([7, 2, 9, 5, 11], dtype=torch.int32)
    slots = (pages[:, None] * ps + torch.arange(ps, dtype=torch.int32)[None, :]).reshape(-1)
    pool.write_tokens(1, slice(0, 20), slots)
    assert pool.req_to_page[1, :5].tolist() == [7, 2, 9, 5, 11]
    assert torch.equal(pool.page_table(torch.tensor([1]), 5)[0], pool.req_to_page[1, :5])
    with pytest.raises(AssertionError):
        pool.write_tokens(1, slice(2, 8), slots[:6])
    # scatter of scattered positions: two requests, one continuing inside a page, one starting a page
    pool.scatter_tokens(torch.tensor([1, 2]), torch.tensor([20, 8]), torch.tensor([3 * ps, 6 * ps], dtype=torch.int32))
    pool.scatter_tokens(torch.tensor([1, 2]), torch.tensor([21, 9]), torch.tensor([3 * ps + 1, 6 * ps + 1], dtype=torch.int32))
    assert pool.req_to_page[1, 5] == 3 and pool.req_to_page[2, 2] == 6 and pool.req_to_token[1, 21] == 13
    one = ReqToTokenPool(2, 8, 1, "cpu")
    assert one.req_to_page is one.req_to_token
