"""Fused lm_head product + greedy reduction (msgl_lm_head_argmax) through the C ABI, against the host oracle of the
reference's two steps (logits_processor.py:56-59: bf16 logits = hidden @ head^T; sampler.py:14-15: argmax of the fp32
view).  Index work is exact wherever it is decidable: the kernel's sums differ from another GEMM's in the last fp32
bits, so a logit may round to the neighbouring bf16 value; the token must equal the oracle's whenever the oracle's
top two logits are more than two bf16 ulps apart, and must always be a column whose exact logit is within one ulp of
the best."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from mini_sglang_amd import ops
    return ops


def _inputs(bs, vocab, hidden, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(bs, hidden, generator=g) * scale).to(torch.bfloat16)
    w = (torch.randn(vocab, hidden, generator=g) * 0.05).to(torch.bfloat16)
    return x, w


def _oracle(x, w):
    """fp64 logits of the bf16 inputs, the bf16 logits the reference's matmul stores (one rounding of the exact sum, up to
    its own summation error), and the greedy token of those."""
    exact = x.double() @ w.double().t()
    stored = exact.float().to(torch.bfloat16)
    return exact, stored, stored.float().argmax(dim=-1)


@pytest.mark.parametrize("bs,vocab,hidden", [(1, 1000, 256), (3, 4099, 512), (33, 2048, 1024), (64, 7777, 1024), (100, 18992, 1024),
                                             (128, 3000, 2560), (129, 1500, 512), (256, 5000, 4096)])
def test_fused_head_matches_oracle(bs, vocab, hidden):
    ops = _ops()
    x, w = _inputs(bs, vocab, hidden, 1000 + bs)
    exact, stored, want = _oracle(x, w)
    idx, val = ops.lm_head_argmax(x.to(DEV), w.to(DEV))
    idx, val = idx.cpu().long(), val.cpu()
    assert int(idx.min()) >= 0 and int(idx.max()) < vocab
    best = exact.max(dim=-1).values
    ulp = torch.exp2(torch.floor(torch.log2(best.abs().clamp_min(1e-30))) - 7)
    chosen = exact.gather(1, idx.view(-1, 1))[:, 0]
    assert bool((chosen >= best - ulp).all()), "a chosen column is more than one bf16 ulp below the best logit"
    # the returned value is the chosen column's logit after one rounding to bf16 (within one ulp of the exact sum's rounding)
    assert bool(((val.double() - chosen).abs() <= ulp).all())
    assert bool((val == val.to(torch.bfloat16).float()).all()), "values must be bf16-representable"
    top2 = exact.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 2 * ulp
    assert bool((idx[clear] == want[clear]).all())
    assert float(clear.float().mean()) > 0.5
    # and against the unfused path on the same device, same rule
    lg = torch.nn.functional.linear(x.to(DEV), w.to(DEV))
    idx2 = ops.argmax(lg).cpu().long()
    assert bool((idx[clear] == idx2[clear]).all())


def test_fused_head_ties_go_to_the_lowest_row_and_nan_counts_as_largest():
    ops = _ops()
    x, w = _inputs(40, 3000, 512, 7)
    w[2500] = w[17]          # identical rows: identical sums whatever lane computes them
    w[900] = w[17]
    x2 = x.clone()
    x2[:] = w[17].float().sign().to(torch.bfloat16) * 4   # every column strongly favours row 17 and its copies
    idx, val = ops.lm_head_argmax(x2.to(DEV), w.to(DEV))
    assert idx.cpu().tolist() == [17] * 40
    wn = w.clone()
    wn[1234, 5] = float("nan")
    wn[2999, 0] = float("nan")
    idx, val = ops.lm_head_argmax(x.to(DEV), wn.to(DEV))
    assert idx.cpu().tolist() == [1234] * 40 and bool(torch.isnan(val).all())
    xn = x.clone()
    xn[3, 100] = float("nan")   # a NaN in one hidden row makes every logit of that column NaN: first row wins
    idx, _ = ops.lm_head_argmax(xn.to(DEV), w.to(DEV))
    assert int(idx[3]) == 0


def test_fused_head_strided_rows_and_refusals():
    ops = _ops()
    from mini_sglang_amd import _capi
    x, w = _inputs(5, 777, 512, 3)
    xs = torch.zeros(5, 1024, dtype=torch.bfloat16)
    xs[:, :512] = x
    ws = torch.zeros(777, 768, dtype=torch.bfloat16)
    ws[:, :512] = w
    a, av = ops.lm_head_argmax(x.to(DEV), w.to(DEV))
    b, bv = ops.lm_head_argmax(xs.to(DEV)[:, :512], ws.to(DEV)[:, :512])
    assert torch.equal(a, b) and torch.equal(av, bv)
    with pytest.raises(RuntimeError):
        ops.lm_head_argmax(x.to(DEV)[:, :384], w.to(DEV)[:, :384])          # hidden not a multiple of 256
    with pytest.raises(RuntimeError):
        ops.lm_head_argmax(x.to(DEV), w.to(DEV), workspace=torch.empty(16, dtype=torch.uint8, device=DEV))
    assert _capi.lib.msgl_lm_head_argmax_workspace_bytes(5, 777) == 7 * 5 * 8


def test_fused_head_full_vocabulary_successor_rows():
    """The benchmark's head size (151 936 x 4096) at batch 256: every column has one planted winner with a wide margin."""
    ops = _ops()
    torch.manual_seed(0)
    vocab, hidden, bs = 151936, 4096, 256
    w = (torch.randn(vocab, hidden, device=DEV) * 0.02).to(torch.bfloat16)
    want = torch.randint(0, vocab, (bs,), device=DEV)
    want[0], want[1] = 0, vocab - 1
    x = (w[want].float() * 8).to(torch.bfloat16)
    idx, val = ops.lm_head_argmax(x, w)
    assert torch.equal(idx.long(), want)
    ref = ops.argmax(torch.nn.functional.linear(x, w), return_values=True)
    assert torch.equal(ref[0], idx)
    assert bool(((val - ref[1]).abs() <= ref[1].abs() * 2 ** -7).all())


def test_stream_read_yardstick_runs_in_every_form_and_refuses_bad_arguments():
    """The benchmark's read-only yardstick (msgl_stream_read / _cfg / _span): every form reads a buffer without touching it
    or the sink, and arguments outside the documented ranges are refused."""
    ops = _ops()
    buf = torch.randint(0, 2 ** 31 - 1, (1 << 20,), dtype=torch.int32, device=DEV)   # 4 MiB
    keep = buf.clone()
    sink = torch.zeros(64, dtype=torch.int32, device=DEV)
    for unroll in (1, 2, 4, 8):
        ops.stream_read(buf, sink, unroll, 16)
        for span in (4, 64, 1024):
            ops.stream_read(buf, sink, unroll, 8, span)
    ops.stream_read(buf[: 1000 * 4 + 4], sink, 4, 8, 64)   # a length that is no multiple of a span or of a step
    torch.cuda.synchronize()
    assert torch.equal(buf, keep) and int(sink.abs().sum()) == 0
    for bad in ((3, 16, 0), (4, 0, 0), (4, 65, 0), (4, 8, 6), (4, 8, 2)):
        with pytest.raises(RuntimeError):
            ops.stream_read(buf, sink, *bad)


def test_fused_head_seeded_random_shapes():
    """Forty seeded shapes (batch 1-200, 1-6000 rows, rows of 256-2048 elements, input scales over two decades): the same
    decidability rule as above on every one of them."""
    import random
    ops = _ops()
    rng = random.Random(20260)
    for case in range(40):
        bs = rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 200])
        vocab = rng.choice([1, 5, 31, 32, 33, 127, 128, 129, 1000, 2049, 6000])
        hidden = rng.choice([256, 512, 768, 1024, 2048])
        x, w = _inputs(bs, vocab, hidden, 5000 + case, scale=rng.choice([0.1, 1.0, 10.0]))
        exact, _, want = _oracle(x, w)
        idx, val = ops.lm_head_argmax(x.to(DEV), w.to(DEV))
        idx, val = idx.cpu().long(), val.cpu()
        assert int(idx.min()) >= 0 and int(idx.max()) < vocab, (case, bs, vocab, hidden)
        best = exact.max(dim=-1).values
        ulp = torch.exp2(torch.floor(torch.log2(best.abs().clamp_min(1e-30))) - 7)
        chosen = exact.gather(1, idx.view(-1, 1))[:, 0]
        assert bool((chosen >= best - ulp).all()), (case, bs, vocab, hidden)
        assert bool(((val.double() - chosen).abs() <= ulp).all()), (case, bs, vocab, hidden)
        if vocab > 1:
            top2 = exact.topk(2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 2 * ulp
            assert bool((idx[clear] == want[clear]).all()), (case, bs, vocab, hidden)
