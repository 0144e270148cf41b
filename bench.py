#!/usr/bin/env python3
"""Headline benchmark: paged decode attention on MI355X, Qwen3-8B geometry, page_size 16, batch 256,
8192-token contexts (the configuration the headline metric is quoted on).

A "step" is one decode-attention call (one transformer layer's worth: new K/V already in the pool, one
query token per request attends to its whole paged context).  Steps cycle over several layers' pools so
that no step can be served from the 256 MiB Infinity Cache.  Inputs are resident in HBM before the timed
region.  value = algorithmic KV bytes (2 * Hkv * 128 * 2 B * sum of sequence lengths, no de-duplication)
per second, summed over ranks.

N > 1 (launched by torch.distributed.run, one rank per GPU): tensor parallel by KV head exactly as the
reference shards attention (Hq/N query heads and Hkv/N kv heads per rank, identical page tables on every
rank, no collective inside attention), so the total work is fixed: "strong" scaling.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
RANKS_PER_DEVICE = 1    # set in main(): > 1 only when ranks share a GPU (rehearsal)


def trace(msg):
    """Progress line on stderr (MSGL_BENCH_TRACE=1): where a long run is, with seconds since start."""
    if os.environ.get("MSGL_BENCH_TRACE"):
        print(f"[bench {time.perf_counter() - _T0:7.1f}s rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def spread(samples, digits=4):
    """Median and range of a list of timings: {"median": ., "min": ., "max": ., "n": .}."""
    v = sorted(samples)
    n = len(v)
    med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
    return {"median": round(med, digits), "min": round(v[0], digits), "max": round(v[-1], digits), "n": n}


def time_graph(f, launches, reps=7, warm=2):
    """Per-launch device milliseconds of f(i), i = 0 .. launches-1, captured in one hipGraph (as a serving loop replays
    its launches: a kernel of a few microseconds is otherwise timed together with the host's launch gaps) and replayed
    `reps` times, each replay between its own pair of HIP events on the replay stream.  Returns the list of reps
    per-launch times."""
    for i in range(min(launches, 4)):
        f(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(launches):
            f(i)
    for _ in range(warm):
        g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / launches)
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--bs", type=int, default=256)
    ap.add_argument("--ctx", type=int, default=8192)
    ap.add_argument("--page-size", type=int, default=16)
    ap.add_argument("--layers", type=int, default=4, help="distinct per-layer pools cycled through")
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--layout", default="interleaved", choices=["interleaved", "split"])
    ap.add_argument("--max-splits", type=int, default=0, help="0 = backend default for the batch size")
    ap.add_argument("--chunk", type=int, default=0, help="0 = backend default")
    ap.add_argument("--graph-grid", action="store_true", help="launch the full bs x max_splits grid a captured graph uses")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-e2e", action="store_true", help="skip the full-model decode tok/s measurement")
    ap.add_argument("--no-extend", action="store_true", help="skip the prefill-attention TFLOP/s measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary decode-attention configurations")
    ap.add_argument("--no-layers", action="store_true", help="skip the layer-kernel (norm / activation / rotary / store) measurements")
    ap.add_argument("--secondary-deadline", type=float, default=420.0,
                    help="seconds the measurements after the headline may take before the line is printed without them")
    ap.add_argument("--e2e-ctx", type=int, default=0, help="0 = 8192 if the whole-model KV fits in HBM, else the largest power of two that does")
    ap.add_argument("--e2e-steps", type=int, default=20, help="timed whole-model decode steps (5 samples of a fifth each)")
    ap.add_argument("--e2e-eager", action="store_true", help="do not capture decode steps in hipGraphs")
    ap.add_argument("--mixed-bs", type=int, default=256, help="largest batch the mixed-length whole-model measurement tries")
    ap.add_argument("--allreduce-trial-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend under N > 1 (gloo: ranks may share a GPU, for rehearsals)")
    return ap.parse_args()


def main():
    args = parse()
    if os.environ.get("MSGL_BENCH_TRACE"):  # a run that stops making progress says where, every two minutes
        import faulthandler

        faulthandler.dump_traceback_later(120, repeat=True, file=sys.stderr)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.allreduce_trial_child:
        return allreduce_trial_child(world, rank, local_rank)
    if world > 1:
        import torch.distributed as dist

        local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    from mini_sglang_amd import ops
    from mini_sglang_amd.attn.hip_attn_backend import splits_for_batch
    from mini_sglang_amd.runtime import synth, workload

    D = 128
    assert args.hkv % world == 0, "tensor parallel degree must divide the kv heads"
    Hq, Hkv = args.hq // world, args.hkv // world
    bs, S, ps, L = args.bs, args.ctx, args.page_size, args.layers
    pages_per_req = -(-S // ps)
    pool_pages = bs * pages_per_req + 64
    slots = (pool_pages + 1) * ps
    pt = workload.make_page_table([S] * bs, ps, pool_pages, seed=3, scatter=True)
    r2t = pt.req_to_token.to(dev)
    seq_lens = pt.seq_lens.to(dev)
    req_idx = torch.arange(bs, dtype=torch.int32, device=dev)
    pools = []
    for layer in range(L):
        if args.layout == "interleaved":
            kv = synth.normal((slots, 2, Hkv, D), 100 + layer * 7 + rank, device=dev)
            pools.append((kv[:, 0], kv[:, 1]))
        else:
            pools.append((synth.normal((slots, Hkv, D), 100 + layer * 7 + rank, device=dev),
                          synth.normal((slots, Hkv, D), 500 + layer * 7 + rank, device=dev)))
    qs = [synth.normal((bs, Hq, D), 900 + layer + rank, device=dev) for layer in range(L)]
    out = torch.empty((bs, Hq, D), dtype=torch.bfloat16, device=dev)
    max_splits = args.max_splits or splits_for_batch(bs)
    chunk = args.chunk or ops.pick_split(S, S * bs, bs, max_splits, Hkv)
    if not args.graph_grid:
        # an eager call knows the longest context on the host and launches only the splits it needs, exactly as
        # HipAttnBackend.init_forward_metadata does; --graph-grid keeps the length-independent grid of a captured step
        max_splits = max(1, min(max_splits, -(-S // chunk)))
    ws = torch.empty(max(ops.decode_workspace_bytes(bs, Hq, D, max_splits), 16), dtype=torch.uint8, device=dev)
    scale = D ** -0.5

    def step(i: int):
        kc, vc = pools[i % L]
        ops.paged_decode(qs[i % L], kc, vc, r2t, req_idx, seq_lens, scale, ps, out=out, workspace=ws,
                         max_splits=max_splits, chunk_tokens=chunk)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    # HIP events on the launch stream (torch's current stream) around every timed step: the per-launch device time is
    # the MEDIAN of the K intervals; `value` stays the wall clock over exactly K steps between two barriers
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step(i)
        evs[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    per_step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    dev_ms_median = per_step_ms[len(per_step_ms) // 2] if len(per_step_ms) % 2 else 0.5 * (per_step_ms[len(per_step_ms) // 2 - 1] + per_step_ms[len(per_step_ms) // 2])
    dev_ms_mean = evs[0].elapsed_time(evs[-1]) / args.steps
    dev_ms_median_rank = dev_ms_median
    if world > 1:
        import torch.distributed as dist

        from mini_sglang_amd.comm import agree

        elapsed, dev_ms_median, dev_ms_mean = (agree(v, "max", dev) for v in (elapsed, dev_ms_median, dev_ms_mean))

    dist_info = None
    if world > 1:
        dist_info = describe_process_group(dev, dev_ms_median_rank)
    bytes_per_step_rank = 2 * Hkv * D * 2 * S * bs
    total_bytes = bytes_per_step_rank * world * args.steps
    value = total_bytes / elapsed / 1e9
    launch_ms = dev_ms_median
    achieved = bytes_per_step_rank / (launch_ms * 1e-3) / 1e9  # per GPU, median device time per launch

    result = {
        "metric": "paged_decode_attention_hbm_bandwidth",
        "value": round(value, 1),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"Qwen3-8B decode attention, one layer per step, batch {bs} x {S} ctx, page_size {ps}, "
                        f"scattered pages, {L} layer pools cycled, TP={world} ({Hq} q / {Hkv} kv heads per GPU)",
            "batch": bs, "ctx": S, "page_size": ps, "hq_per_gpu": Hq, "hkv_per_gpu": Hkv, "head_dim": D,
            "kv_layout": args.layout, "max_splits": max_splits, "chunk_tokens": chunk, "parallelism": f"tp{world}",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": None,
            "kernel": "paged_decode_kernel (+ split merge)",
            "launch_ms": round(launch_ms, 4),
            "launch_ms_stat": f"median of {args.steps} HIP-event intervals on the launch stream",
            "launch_ms_mean": round(dev_ms_mean, 4),
            "launch_ms_min_max": [round(per_step_ms[0], 4), round(per_step_ms[-1], 4)],
            # the same two figures from the MEAN over the timed steps (earlier records of this benchmark quoted the mean)
            "achieved_from_mean": round(bytes_per_step_rank / (dev_ms_mean * 1e-3) / 1e9, 1),
            "frac_from_mean": round(bytes_per_step_rank / (dev_ms_mean * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "bytes_per_launch": bytes_per_step_rank,
        },
    }

    if dist_info is not None:
        result["distributed"] = dist_info
    # an on-box yardstick beside the spec peak: this library's read-only stream kernel (16 B per lane, non-temporal,
    # nothing stored) over the first layer's pool, which is far larger than the caches; its rate depends on the loads
    # in flight per lane and the resident work groups, so the best of a small sweep is quoted (earlier records of this
    # benchmark quoted a device-to-device copy here, then the single configuration (8, 16): see yardstick_changed_from)
    try:
        buf = pools[0][0] if args.layout == "split" else pools[0][0]._base
        buf = buf.view(-1)
        nbytes = buf.numel() * buf.element_size()
        sink = torch.zeros(64, dtype=torch.int32, device=dev)
        best, sweep = None, {}
        # (unroll, work groups per CU, span KiB): span 0 = grid-stride over the whole buffer, else one work group per
        # contiguous span, the order in which a gather of whole pages walks memory
        for unroll, gpc, span in ((8, 16, 0), (8, 8, 0), (4, 8, 0), (4, 16, 0), (4, 32, 0), (2, 16, 0), (2, 32, 0), (1, 32, 0),
                                  (4, 8, 64), (4, 16, 64), (8, 8, 64), (4, 8, 256), (4, 16, 256), (8, 8, 1024), (4, 16, 1024)):
            for _ in range(2):
                ops.stream_read(buf, sink, unroll, gpc, span)
            c = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            c[0].record()
            for i in range(5):
                ops.stream_read(buf, sink, unroll, gpc, span)
                c[i + 1].record()
            torch.cuda.synchronize()
            ts = sorted(c[i].elapsed_time(c[i + 1]) for i in range(5))
            gbps = nbytes / (ts[2] * 1e-3) / 1e9
            sweep[f"unroll{unroll}_groups{gpc}" + (f"_span{span}k" if span else "")] = round(gbps, 1)
            if best is None or gbps > best[0]:
                best = (gbps, unroll, gpc, span)
        read_gbps = best[0]
        result["roofline"]["measured_stream_read_GBps"] = round(read_gbps, 1)
        result["roofline"]["stream_read_config"] = {"loads_in_flight_per_lane": best[1], "work_groups_per_cu": best[2],
                                                    "span_KiB": best[3], "sweep_GBps": sweep}
        result["roofline"]["stream_read_bytes"] = nbytes
        result["roofline"]["frac_of_measured_stream_read"] = round(achieved / read_gbps, 4)
        result["roofline"]["yardstick_changed_from"] = "r01: device-to-device copy (read + write); r02: read-only stream, one configuration; r03: best of a sweep over loads in flight, resident groups and grid-stride / span order"
    except Exception as e:
        result["roofline"]["measured_stream_read_error"] = f"{type(e).__name__}: {e}"[:120]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(pools[0], qs[0], r2t, S, ps, Hq, Hkv, D, scale, args.cpu_seconds)
    # HBM bytes per launch of this exact shape from rocprofv3 PMC passes, as scripts/pmc_decode.sh collected them and
    # scripts/pmc_traffic.py reduced them to profiles/decode_traffic.json (the corrections the counters need on gfx950
    # are applied there and named in the file); null for any other shape or when the file is absent
    result["roofline"].update(recorded_traffic(args.layout, bs, S, ps, Hq, Hkv))
    del pools, qs, ws, out
    torch.cuda.empty_cache()
    # The secondary measurements below must never cost the headline line: if one of them is still running
    # after the deadline (a collective that never completes under TP, say), every rank gives up together and
    # rank 0 prints what it has.
    printed = threading.Event()

    def emit():
        if rank == 0 and not printed.is_set():
            printed.set()
            print(json.dumps(result), flush=True)

    def give_up():
        result["secondary_timeout_s"] = args.secondary_deadline
        emit()
        os._exit(0)

    watchdog = threading.Timer(args.secondary_deadline, give_up)
    watchdog.daemon = True
    watchdog.start()
    trace("headline done")
    if not args.no_configs:
        result["other_configs"] = other_decode_configs(args, world, rank)
        trace("other_configs done")
    if not args.no_extend:
        result["extend"] = extend_attention(args, world, rank)
        trace("extend done")
    if not args.no_layers:
        result["layers"] = layer_kernels(args, world, rank)
        trace("layers done")
    xgmi_ok = False
    # ranks per physical GPU: 1 on a node with a GPU per rank, more in a one-GPU rehearsal, where every rank must
    # size its pools for its share of the memory it sees free
    global RANKS_PER_DEVICE
    RANKS_PER_DEVICE = max(1, -(-world // max(1, torch.cuda.device_count())))
    if world > 1 and not args.no_e2e:
        result["tp_allreduce"] = tp_allreduce_trial_isolated(world, rank, local_rank, barrier)
        xgmi_ok = bool(result["tp_allreduce"].get("validated"))
        trace(f"all-reduce trial done, validated={xgmi_ok}")
    rehearsal = world > 1 and args.dist_backend != "nccl"
    if rehearsal and not args.no_e2e:
        # A CPU-side backend (gloo) can neither be captured in a graph nor reduce bf16 tensors on the GPU: the
        # whole-model legs then run eagerly with every hidden-state all-reduce on the direct-read kernels (issued
        # step by step when the ranks share a device).  Same control flow as the RCCL run; the times mean nothing.
        if xgmi_ok:
            from mini_sglang_amd import comm

            args.e2e_eager = True
            comm.enable_xgmi_allreduce()
            result["rehearsal"] = f"{args.dist_backend} backend, eager steps, all-reduce on direct peer reads, {RANKS_PER_DEVICE} rank(s) per GPU"
        else:
            result["decode_e2e"] = {"skipped": f"whole-model decode under {args.dist_backend} needs the direct-read all-reduce, which did not validate"}
            args.no_e2e = True
    if not args.no_e2e:
        result["decode_e2e"] = decode_e2e(args, world, rank, barrier)
        trace("decode_e2e done")
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        result["decode_e2e_mixed"] = decode_e2e_mixed(args, world, rank, barrier)
        trace("decode_e2e_mixed done")
        if world == 1:
            gc.collect()
            torch.cuda.empty_cache()
            result["decode_e2e_small"] = decode_e2e_small(args, barrier)
        if rehearsal:
            from mini_sglang_amd import comm

            comm.disable_xgmi_allreduce()
        elif xgmi_ok:  # the same steps with the o_proj / down_proj / embedding all-reduces on direct peer reads
            from mini_sglang_amd import comm

            try:
                gc.collect()
                torch.cuda.empty_cache()
                comm.enable_xgmi_allreduce()
                again = decode_e2e(args, world, rank, barrier)
                result["decode_e2e"]["xgmi_allreduce"] = {k: again[k] for k in again
                                                          if k.startswith(("tok_per_s", "ms_per_step", "error", "ctx", "hip_graph"))}
            finally:
                comm.disable_xgmi_allreduce()
    watchdog.cancel()
    emit()
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


def describe_process_group(dev, launch_ms_rank):
    """What lets a reader check that an N > 1 line came from N ranks on N distinct GPUs: the process group's own
    world size and backend, the RCCL version torch was built against, and, gathered over the ranks, every rank's
    device identity (host / GPU UUID) and its own per-launch time."""
    import torch.distributed as dist

    from mini_sglang_amd.comm import tp_all_gather
    from mini_sglang_amd.comm.xgmi_allreduce import _device_identity as device_identity

    info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rank_launch_ms": None, "devices": None}
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:  # noqa: BLE001
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    ident = device_identity(dev).encode()[:96].ljust(96, b"\0")
    mine = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
    info["devices"] = [bytes(t.cpu().tolist()).rstrip(b"\0").decode("utf-8", "replace") for t in tp_all_gather(mine)]
    info["distinct_devices"] = len(set(info["devices"]))
    ms = torch.tensor([launch_ms_rank], dtype=torch.float64, device=dev).view(torch.uint8)
    info["rank_launch_ms"] = [round(float(t.view(torch.float64)[0]), 4) for t in tp_all_gather(ms)]
    return info


def recorded_traffic(layout, bs, S, ps, Hq, Hkv):
    path = os.path.join(ROOT, "profiles", "decode_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return {"traffic": None}
    shape = rec.get("shape", {})
    if shape != {"layout": layout, "batch": bs, "ctx": S, "page_size": ps, "hq": Hq, "hkv": Hkv}:
        return {"traffic": None}
    return {"traffic": rec.get("traffic_bytes_per_launch"), "traffic_source": "profiles/decode_traffic.json: " + rec.get("how", ""),
            "traffic_measured_at_commit": rec.get("commit")}


def recorded_extend_counters(prompt, prefix, ps, Hq, Hkv):
    """Matrix-core busy fraction of the extend launch from the committed profiler record (profiles/extend_counters.json,
    written by the profiler script under scripts/), when it was taken on this shape; counters cannot be read inside a timed run."""
    try:
        with open(os.path.join(ROOT, "profiles", "extend_counters.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return {}
    if rec.get("shape") != {"prompt": prompt, "prefix": prefix, "page_size": ps, "hq": Hq, "hkv": Hkv}:
        return {}
    return {"matrix_core_busy": rec.get("matrix_core_busy"), "matrix_core_busy_source": "profiles/extend_counters.json: " + rec.get("how", ""),
            "matrix_core_busy_measured_at_commit": rec.get("commit"), "kernel": rec.get("kernel")}


def mixed_radix_batch(bs=256):
    """The mixed configuration of the benchmark family: `bs` contexts log-uniform in 1k-32k tokens (seed 3), groups of 16
    requests on a shared page-aligned radix prefix of 1-4k tokens.  Returns (lens, groups, prefix_lens)."""
    from mini_sglang_amd.runtime import workload

    lens = workload.mixed_lengths(bs, 1024, 32768, 3)
    groups = [i // 16 for i in range(bs)]
    prefix = [1024 * (1 + g % 4) for g in range(bs // 16)]
    lens = [max(n, prefix[g] + 16) for n, g in zip(lens, groups)]
    return lens, groups, prefix


def build_decode_config(Hq, Hkv, ps, lens, groups=None, prefix=None, layers=2, share_once=False, dev=None):
    """One decode-attention configuration as HipAttnBackend launches it per layer: page table, per-layer pools of random
    K/V, the split size / launch plan / shared-prefix arrays the backend would choose, and launch(i) = the call(s) of
    layer i % layers.  The timed legs and the parity tests of the same shapes both go through this."""
    from mini_sglang_amd import ops
    from mini_sglang_amd.attn.hip_attn_backend import splits_for_batch
    from mini_sglang_amd.runtime import workload

    D = 128
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    bs = len(lens)
    pages = sum(-(-n // ps) for n in lens) + 8
    pt = workload.make_page_table(lens, ps, pages, 3, groups=groups, group_prefix_lens=prefix)
    slots = (pages + 1) * ps
    pools = [torch.empty((slots, 2, Hkv, D), dtype=torch.bfloat16, device=dev).normal_() for _ in range(layers)]
    q = torch.randn((bs, Hq, D), device=dev).bfloat16()
    o = torch.empty_like(q)
    r2t, sl = pt.req_to_token.to(dev), pt.seq_lens.to(dev)
    idx = torch.arange(bs, dtype=torch.int32, device=dev)
    ms = splits_for_batch(bs)
    # what the backend does per step: ragged batches get the split size of the schedule model and a launch plan
    ragged = max(lens) * bs > 1.25 * sum(lens)
    chunk = ops.pick_split_planned(lens, ms, Hkv) if ragged else ops.pick_split(max(lens), sum(lens), bs, ms, Hkv)
    ms = max(1, min(ms, -(-max(lens) // chunk)))
    plan = ops.decode_plan(sl, ms, chunk) if ragged and ms > 1 else None  # once per step, shared by all layers
    ws = torch.empty(max(16, ops.decode_workspace_bytes(bs, Hq, D, ms)), dtype=torch.uint8, device=dev)
    reuse = torch.tensor([prefix[g] for g in groups], dtype=torch.int32, device=dev) if groups is not None else None

    cascade = None
    if groups is not None and share_once:
        # the backend's eager path for such a batch: every group's prefix once through the matrix-core kernel
        # (msgl_shared_prefix_decode), the private tails per request, one merge
        ng = max(groups) + 1
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
        first, count = [groups.index(g) for g in range(ng)], [groups.count(g) for g in range(ng)]
        key_chunk = 1024
        start, extra = i32([prefix[g] for g in groups]), i32([-(-prefix[g] // key_chunk) for g in groups])
        rest = [n - prefix[g] for n, g in zip(lens, groups)]
        most_extra = max(-(-p_ // key_chunk) for p_ in prefix)
        chunk = ops.pick_split_planned(rest, splits_for_batch(bs) - most_extra, Hkv)
        ms = -(-max(rest) // chunk) + most_extra
        ws = torch.empty(ops.decode_workspace_bytes(bs, Hq, D, ms), dtype=torch.uint8, device=dev)
        plan = ops.decode_plan(sl, ms, chunk, start_lens=start)
        cascade = (i32(first), i32(count), i32(prefix), max(count), max(prefix), key_chunk, start, extra)

    def launch(i):
        kv = pools[i % layers]
        if cascade is not None:
            gf, gc, gp, mm, mp, kc, start, extra = cascade
            ops.shared_prefix_decode(q, kv[:, 0], kv[:, 1], r2t, idx, sl, start, gf, gc, gp, mm, mp, kc, D ** -0.5, ws, ms, chunk)
            ops.paged_decode(q, kv[:, 0], kv[:, 1], r2t, idx, sl, D ** -0.5, ps, out=o, workspace=ws, max_splits=ms,
                             chunk_tokens=chunk, plan=plan, start_lens=start, extra_splits=extra)
            return
        ops.paged_decode(q, kv[:, 0], kv[:, 1], r2t, idx, sl, D ** -0.5, ps, out=o, workspace=ws, max_splits=ms,
                         chunk_tokens=chunk, reuse_lens=reuse, plan=plan)

    return {"launch": launch, "q": q, "out": o, "pools": pools, "page_table": pt, "req_to_token": r2t, "seq_lens": sl,
            "req_idx": idx, "chunk": chunk, "max_splits": ms, "plan": plan, "cascade": cascade, "workspace": ws, "reuse_lens": reuse}


def other_decode_configs(args, world, rank):
    """Decode attention (one layer per launch, same accounting as the headline: algorithmic KV bytes, no
    de-duplication of shared prefixes) on the other single-layer shapes the benchmark family names: Qwen3-0.6B
    batch 64 x 2k at page_size 1, and Qwen3-8B batch 256 with mixed 1k-32k contexts where groups of 16 requests
    share a page-aligned 1-4k radix prefix.  Per-GPU figures under TP (heads divided like the headline)."""
    from mini_sglang_amd import ops
    from mini_sglang_amd.attn.hip_attn_backend import splits_for_batch
    from mini_sglang_amd.runtime import workload

    dev = torch.device("cuda", torch.cuda.current_device())
    D = 128
    out = {}

    def run(name, Hq, Hkv, ps, lens, groups=None, prefix=None, layers=2, steps=12, share_once=False):
        cfg = build_decode_config(Hq, Hkv, ps, lens, groups, prefix, layers, share_once, dev)
        f, pt, chunk, plan, cascade = cfg["launch"], cfg["page_table"], cfg["chunk"], cfg["plan"], cfg["cascade"]
        # the launches of one timing pass are captured in a hipGraph, as a serving loop replays them (a 20-90 us kernel
        # is otherwise timed together with the host's launch gaps); 7 replays, each between its own events: median
        per = time_graph(f, steps, reps=7)
        sp = spread([x * 1e3 for x in per], 2)  # microseconds per launch
        t = sp["median"] * 1e-6
        b = 2 * Hkv * D * 2
        algo, uniq = b * sum(lens) / t / 1e9, b * pt.unique_tokens / t / 1e9
        out[name] = {"GBps_per_gpu": round(algo, 1), "unique_GBps_per_gpu": round(uniq, 1),
                     "us_per_launch": sp, "ms_per_launch": round(t * 1e3, 4),
                     # the roofline fraction is quoted on the bytes the kernel must move at least once (unique slots):
                     # with radix-shared prefixes the algorithmic count (every request's whole context) exceeds them
                     "frac_of_hbm_peak": round(uniq / HBM_PEAK_GBPS, 4),
                     "algorithmic_GBps_over_peak": round(algo / HBM_PEAK_GBPS, 4),
                     "chunk_tokens": chunk, "planned_launch": plan is not None, "shared_prefix_once": cascade is not None}

    try:
        if world == 1:
            run("qwen3_0.6b_bs64_ctx2048_ps1", 16, 8, 1, [2048] * 64, layers=8)
            # the long-context configuration: one TP-8 shard of Qwen3-32B (8 q / 1 kv head), page_size 64, 128k contexts
            run("qwen3_32b_tp8_shard_bs1_ctx131072_ps64", 8, 1, 64, [131072], layers=8, steps=20)
            run("qwen3_32b_tp8_shard_bs8_ctx131072_ps64", 8, 1, 64, [131072] * 8, layers=4, steps=12)
        lens, groups, prefix = mixed_radix_batch(256)
        run("qwen3_8b_bs256_mixed_1k_32k_ps16_radix_shared", args.hq // world, args.hkv // world, 16, lens, groups, prefix)
        run("qwen3_8b_bs256_mixed_1k_32k_ps16_radix_shared_prefix_once", args.hq // world, args.hkv // world, 16, lens, groups, prefix,
            share_once=True)
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    torch.cuda.empty_cache()
    return out


def build_extend_config(Hq, Hkv, ps, Sq, prefix, seed=31, dev=None):
    """One prefill-attention launch as HipAttnBackend.forward_extend issues it: a single request of `Sq` new tokens on a
    `prefix`-token cached prefix, scattered pages, and -- when the launch has fewer work items (128-row query tiles x
    head pairs) than compute units over a long key range -- the workspace that lets the library cut the items' keys
    into runs.  The timed legs and the parity tests of the same shapes both go through this."""
    from mini_sglang_amd import ops
    from mini_sglang_amd.runtime import synth, workload

    D = 128
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    S = prefix + Sq
    pages = -(-S // ps) + 8
    pt = workload.make_page_table([S], ps, pages, seed=5, scatter=True)
    kv = synth.normal(((pages + 1) * ps, 2, Hkv, D), seed, device=dev)
    q = synth.normal((Sq, Hq, D), seed + 1, device=dev)
    out = torch.empty_like(q)
    r2t = pt.req_to_token.to(dev)
    idx = torch.zeros(1, dtype=torch.int32, device=dev)
    cu = torch.tensor([0, Sq], dtype=torch.int32, device=dev)
    pre = torch.tensor([prefix], dtype=torch.int32, device=dev)
    ks = ops.extend_key_splits(1, Sq, S, Hq, Hkv)
    ws = torch.empty(ops.extend_workspace_bytes_for(Sq, Hq, D, ks), dtype=torch.uint8, device=dev) if ks > 1 else None

    def launch():
        ops.paged_extend(q, kv[:, 0], kv[:, 1], r2t, idx, cu, pre, Sq, D ** -0.5, ps, out=out, max_seqlen_k=S, workspace=ws)

    return {"launch": launch, "q": q, "kv": kv, "out": out, "req_to_token": r2t, "page_table": pt, "workspace": ws}


def extend_attention(args, world, rank):
    """Prefill attention of one transformer layer (the MFMA-bound half of the path): one 16384-token prompt, the
    reference's default prefill budget, no cached prefix, same heads and page size as the headline shape.  Reported
    beside the headline number, never as `value`: causal flops / HIP-event time per launch on this rank."""
    from mini_sglang_amd import ops
    from mini_sglang_amd.runtime import synth, workload

    D = 128
    dev = torch.device("cuda", torch.cuda.current_device())

    def measure(Hq, Hkv, ps, Sq, prefix, steps=10):
        cfg = build_extend_config(Hq, Hkv, ps, Sq, prefix, seed=31 + rank, dev=dev)
        f, ws = cfg["launch"], cfg["workspace"]
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        evs[0].record()
        for i in range(steps):
            f()
            evs[i + 1].record()
        torch.cuda.synchronize()
        sp = spread([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
        ms = sp["median"]
        tf = 4 * Hq * D * Sq * (prefix + (Sq + 1) / 2) / (ms * 1e-3) / 1e12
        what = f"one {Sq}-token prompt" + (f" on a {prefix}-token cached prefix" if prefix else "")
        return {"workload": f"{what}, {Hq} q / {Hkv} kv heads per GPU, page_size {ps}", "ms_per_launch": round(ms, 4), "ms_per_launch_spread": sp, "key_split_workspace": ws is not None,
                "TFLOPs_per_gpu": round(tf, 1),
                "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4)}}

    try:
        res = measure(args.hq // world, args.hkv // world, args.page_size, 16384, 0)
        res["roofline"].update(recorded_extend_counters(16384, 0, args.page_size, args.hq // world, args.hkv // world))
        if world == 1:  # the long-context configuration's prefill: one TP-8 shard of Qwen3-32B (8 q / 1 kv head), page_size 64
            res["qwen3_32b_tp8_shard"] = [measure(8, 1, 64, 16384, 0), measure(8, 1, 64, 4096, 65536, steps=6)]
        return res
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def layer_kernels(args, world, rank):
    """The fused layer kernels of the path (RMSNorm plain / fused-add / per-head, SiluAndMul, RoPE, the fused
    q/k-norm + RoPE + KV store, the plain KV store) on Qwen3-8B shapes at prefill (16384 tokens) and decode (256 tokens)
    sizes, this rank's shard under TP.  Per op: algorithmic bytes (every input element read once, every output element
    written once), median device time per launch from 7 replays of a hipGraph of `launches` calls that rotate over
    buffer sets larger than the 256 MiB Infinity Cache (prefill sizes; the decode-size tensors are a few MiB and stay
    cache resident, as they are in a serving step right behind the GEMM that produced them, and their times are launch
    latency, not bandwidth), and the fraction of the HBM peak.  KV slots are whole pages of `--page-size` tokens in
    random order, as the allocator hands them to a prefill.  One op (RMSNorm at the prefill size) also carries the CPU oracle's rate on the host cores."""
    from mini_sglang_amd import ops
    from mini_sglang_amd.layers.rotary_embedding import build_cos_sin_cache

    dev = torch.device("cuda", torch.cuda.current_device())
    hid, inter, Hq, Hkv, D = 4096, 12288 // world, args.hq // world, args.hkv // world, 128
    out = {"shapes": f"Qwen3-8B, TP={world}: hidden {hid}, intermediate {inter}, {Hq} q / {Hkv} kv heads per GPU", "ops": {}}

    def record(name, T, nbytes, f, launches):
        per = time_graph(f, launches, reps=7)
        sp = spread([x * 1e3 for x in per], 2)  # microseconds
        gbps = nbytes / (sp["median"] * 1e-6) / 1e9
        out["ops"][f"{name}_T{T}"] = {"bytes_per_launch": nbytes, "us_per_launch": sp, "GBps": round(gbps, 1),
                                     "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4)}

    try:
        cache = build_cos_sin_cache(D, 40960, 1e6).to(dev)
        w = (torch.randn(hid, device=dev) * 0.2 + 1).bfloat16()
        wd = (torch.randn(D, device=dev) * 0.2 + 1).bfloat16()
        for T in (16384, 256):
            big = T > 1024
            nb = 4 if big else 1                      # buffer sets cycled: 4 x (134 + 134) MB for [16384, 4096] in + out
            launches = 8 if big else 32
            xs = [torch.randn(T, hid, device=dev).bfloat16() for _ in range(nb)]
            ys = [torch.empty_like(x) for x in xs]
            record("rmsnorm", T, 2 * T * hid * 2 + hid * 2, lambda i: ops.rmsnorm(xs[i % nb], w, 1e-6, out=ys[i % nb]), launches)
            record("fused_add_rmsnorm", T, 4 * T * hid * 2 + hid * 2, lambda i: ops.fused_add_rmsnorm(xs[i % nb], ys[i % nb], w, 1e-6), launches)
            if big and rank == 0 and world == 1:
                out["ops"][f"rmsnorm_T{T}"]["cpu_oracle"] = cpu_layer_baseline(xs[0], w)
            del xs, ys
            # per-head q/k norm as a stand-alone call: [T * (Hq + Hkv), 128] rows
            rows = T * (Hq + Hkv)
            hs = [torch.randn(rows, D, device=dev).bfloat16() for _ in range(nb)]
            ho = [torch.empty_like(h) for h in hs]
            record("rmsnorm_heads128", T, 2 * rows * D * 2 + D * 2, lambda i: ops.rmsnorm(hs[i % nb], wd, 1e-6, out=ho[i % nb]), launches)
            del hs, ho
            ng = 2 if big else 1                      # [16384, 2 x 12288] bf16 is 805 MB: two sets suffice
            gs = [torch.randn(T, 2 * inter, device=dev).bfloat16() for _ in range(ng)]
            go = [torch.empty(T, inter, device=dev, dtype=torch.bfloat16) for _ in range(ng)]
            record("silu_and_mul", T, 3 * T * inter * 2, lambda i: ops.silu_and_mul(gs[i % ng], out=go[i % ng]), launches)
            del gs, go
            qkvs = [torch.randn(T, (Hq + 2 * Hkv) * D, device=dev).bfloat16() for _ in range(nb)]
            pos = torch.randint(0, 40960, (T,), device=dev, dtype=torch.int32)

            def rope(i):
                t = qkvs[i % nb]
                ops.rope_neox_(pos, t[:, : Hq * D], t[:, Hq * D: (Hq + Hkv) * D], cache, D)

            record("rope_neox", T, 2 * T * (Hq + Hkv) * D * 2 + T * D * 4, rope, launches)
            # slots as the page allocator hands them to a prefill: whole pages in random order, consecutive slots inside
            ps = args.page_size
            n_pages = -(-T // ps)
            slots = (n_pages + 5) * ps
            kvs = [torch.zeros(slots, 2, Hkv, D, device=dev, dtype=torch.bfloat16) for _ in range(nb)]
            page_of = (torch.randperm(n_pages + 4, device=dev)[:n_pages] + 1).repeat_interleave(ps)[:T]
            loc = (page_of * ps + torch.arange(T, device=dev) % ps).to(torch.int32)

            def fused(i):
                kv = kvs[i % nb]
                ops.qknorm_rope_store_(qkvs[i % nb], wd, wd, 1e-6, pos, cache, Hq, Hkv, D, loc=loc, k_cache=kv[:, 0], v_cache=kv[:, 1])

            # qkv row read once; q and k written back in place; k and v written to the pool; one cos/sin row per token
            record("qknorm_rope_store", T, T * ((Hq + 2 * Hkv) * D * 2 + (Hq + Hkv) * D * 2 + 2 * Hkv * D * 2 + D * 4), fused, launches)

            def store(i):
                t, kv = qkvs[i % nb], kvs[i % nb]
                ops.store_kv(t[:, Hq * D: (Hq + Hkv) * D].view(T, Hkv, D), t[:, (Hq + Hkv) * D:].view(T, Hkv, D), loc, kv[:, 0], kv[:, 1])

            record("store_kv", T, T * (4 * Hkv * D * 2 + 4), store, launches)
            del qkvs, kvs
            torch.cuda.empty_cache()
        # greedy head: product with this rank's vocabulary rows and arg-max in one pass over the rows, beside the
        # unfused pair it replaces (library GEMM writing [bs, vocab / tp] logits, then msgl_argmax_val)
        vocab = 151936 // world
        head = (torch.randn(vocab, hid, device=dev) * 0.02).bfloat16()
        for bs in (32, 64, 128):
            hx = torch.randn(bs, hid, device=dev).bfloat16()
            hws = ops.lm_head_argmax_workspace(bs, vocab, dev)
            record("lm_head_argmax", bs, vocab * hid * 2 + bs * hid * 2, lambda i: ops.lm_head_argmax(hx, head, workspace=hws), 4)
            per = time_graph(lambda i: ops.argmax(torch.nn.functional.linear(hx, head), return_values=True), 4, reps=7)
            out["ops"][f"lm_head_argmax_T{bs}"]["gemm_then_argmax_us"] = spread([x * 1e3 for x in per], 2)
        del head
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    torch.cuda.empty_cache()
    return out


def cpu_layer_baseline(x_dev, w_dev):
    """The oracle's RMSNorm (layernorm.py:39-50 restated) on the host cores over the same [T, hidden] tensor."""
    from oracle import torch_oracle as orc

    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = max(1, min(allowed, omp) if omp > 0 else allowed)
    torch.set_num_threads(cores)
    x, w = x_dev.cpu(), w_dev.cpu()
    orc.rmsnorm_ref(x, w, 1e-6)
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or time.perf_counter() - t0 < 2.0:
        orc.rmsnorm_ref(x, w, 1e-6)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"GBps": round(2 * x.numel() * 2 / dt / 1e9, 2), "ms": round(dt * 1e3, 1), "cores": cores, "kind": "port", "passes": reps}


def decode_e2e_small(args, barrier):
    """Whole-model decode tok/s of the small configuration: Qwen3-0.6B shape, batch 64 x 2048 ctx, page_size 1
    (per-token page table, fully scattered slots), one GPU, hipGraph replay."""
    from mini_sglang_amd.runtime import decode_loop
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    try:
        bs, ctx, ps, samples, each = 64, 2048, 1, 6, 6
        mr, batch = decode_loop.build_decode_state(Qwen3Config.qwen3_0_6b(), bs, ctx, ps, samples * each + 24 + 2 * (5 * 8 + 2), use_graph=not args.e2e_eager)
        sp = spread([dt / each * 1e3 for dt in decode_loop.time_decode_samples(mr, batch, samples, each, 4, barrier=barrier)], 3)
        info = {"model": "Qwen3-0.6B shape, random weights", "batch": bs, "ctx": ctx, "page_size": ps, "tp": 1,
                "tok_per_s": round(bs / (sp["median"] * 1e-3), 1), "ms_per_step": sp["median"], "ms_per_step_spread": sp,
                "steps": samples * each, "timing": f"median of {samples} samples of {each} steps", "hip_graph": not args.e2e_eager}
        if not args.e2e_eager:
            # the same steps chained on the device (step preparation, forward pass and greedy token in one graph, 8 steps
            # per host synchronisation); at this batch size the head is the fused product + arg-max, timed beside
            # the chain with GEMM + argmax instead
            try:
                for key, fused in (("chained", True), ("chained_gemm_argmax", False)):
                    mr.fused_head = fused
                    dts = decode_loop.time_decode_chained_samples(mr, batch, 5, 8, 2, barrier=barrier)
                    spc = spread([dt / 8 * 1e3 for dt in dts], 3)
                    info.update({f"tok_per_s_{key}": round(bs / (spc["median"] * 1e-3), 1), f"ms_per_step_{key}": spc["median"],
                                 f"ms_per_step_{key}_spread": spc})
                info["chain"] = 8
            except Exception as e:  # noqa: BLE001
                info["chained_error"] = f"{type(e).__name__}: {e}"[:200]
            finally:
                mr.fused_head = True
        return info
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def decode_e2e_mixed(args, world, rank, barrier):
    """Whole-model decode tok/s on the mixed configuration: Qwen3-8B shape, contexts log-uniform in 1k-32k, groups of
    16 requests on a shared page-aligned 1-4k radix prefix, page_size 16; the largest batch (256, 128, 64 ...) whose
    K/V fits this GPU's memory beside the weights.  hipGraph replay: the decode kernels run in planned launch order."""
    from mini_sglang_amd.runtime import decode_loop, workload
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    try:
        cfg = Qwen3Config.qwen3_8b()
        ps = 16
        free, _ = torch.cuda.mem_get_info()
        free //= RANKS_PER_DEVICE
        if world > 1:  # every rank must build the same batch: size it for the rank with the least free memory
            import torch.distributed as dist

            from mini_sglang_amd.comm import agree

            free = int(agree(float(free), "min", "cuda"))
        per_token = 2 * (cfg.num_key_value_heads // world) * cfg.head_dim * 2 * cfg.num_hidden_layers
        bs = max(16, args.mixed_bs)
        while True:
            lens = workload.mixed_lengths(bs, 1024, 32768, 3)
            groups = [i // 16 for i in range(bs)]
            prefix = [1024 * (1 + g % 4) for g in range(bs // 16)]
            lens = [max(n, prefix[g] + 16) for n, g in zip(lens, groups)]
            unique = sum(lens) - sum(15 * p for p in prefix)
            if (unique + bs * 96) * per_token + 2 * 8.2e9 / world <= 0.8 * free or bs <= 16:
                break
            bs //= 2
        samples, each = 5, 2
        steps = samples * each
        info = {"model": "Qwen3-8B shape, random weights", "batch": bs, "ctx": "log-uniform 1k-32k, 16 requests per shared 1-4k prefix",
                "tokens_in_context": sum(lens), "page_size": ps, "tp": world, "steps": steps,
                "timing": f"median of {samples} samples of {each} steps", "hip_graph": not args.e2e_eager}
        # every request reads its prefix itself, then: each group's prefix attended once per step (msgl_shared_prefix_decode)
        for suffix, share in (("", False), ("_prefix_once", True)):
            import gc

            mr, batch = decode_loop.build_decode_state(cfg, bs, 0, ps, 2 * steps + 24, use_graph=not args.e2e_eager, seq_lens=lens,
                                                       groups=groups, group_prefix_lens=prefix, share_prefix_in_graphs=share)
            mr.attn_backend.share_prefix_reads = share
            dts = decode_loop.time_decode_samples(mr, batch, samples, each, 3, barrier=barrier)
            if world > 1:
                from mini_sglang_amd.comm import agree

                dts = [agree(dt, "max", "cuda") for dt in dts]
            sp = spread([dt / each * 1e3 for dt in dts], 3)
            info["tok_per_s" + suffix] = round(bs / (sp["median"] * 1e-3), 1)
            info["ms_per_step" + suffix] = sp["median"]
            info["ms_per_step_spread" + suffix] = sp
            del mr, batch
            gc.collect()
            torch.cuda.empty_cache()
        return info
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def decode_e2e(args, world, rank, barrier):
    """Decode tok/s of the whole Qwen3-8B-shaped stack (random weights, random K/V contexts): host batch
    preparation, hipGraph replay of the forward (GEMMs on hipBLASLt, everything else on this repo's kernels,
    RCCL all-reduce under TP), greedy sampling.  Reported beside the headline number, never as `value`."""
    from mini_sglang_amd.runtime import decode_loop
    from mini_sglang_amd.runtime.qwen3 import Qwen3Config

    try:
        cfg = Qwen3Config.qwen3_8b()
        bs, ps = args.bs, args.page_size
        free, total = torch.cuda.mem_get_info()
        free //= RANKS_PER_DEVICE
        weights = 2 * 8.2e9 / world
        per_token = 2 * (cfg.num_key_value_heads // world) * cfg.head_dim * 2 * cfg.num_hidden_layers
        ctx = args.e2e_ctx
        if ctx <= 0:
            ctx = 8192
            while ctx > 256 and bs * (ctx + 160) * per_token + weights > 0.85 * free:
                ctx //= 2
            if world > 1:  # every rank must build the same page tables: agree on the smallest choice
                import torch.distributed as dist

                from mini_sglang_amd.comm import agree

                ctx = int(agree(float(ctx), "min", "cuda"))
        info = {"model": "Qwen3-8B shape, random weights", "batch": bs, "ctx": ctx, "page_size": ps, "tp": world}
        use_graph = not args.e2e_eager
        try:
            mr, batch = decode_loop.build_decode_state(cfg, bs, ctx, ps, args.e2e_steps + 5 * 8 + 40, use_graph=use_graph)
        except Exception as e:  # graph capture refused (e.g. a collective that cannot be captured): run eagerly
            if not use_graph:
                raise
            info["graph_error"] = f"{type(e).__name__}: {e}"[:200]
            use_graph = False
            torch.cuda.synchronize()
            mr, batch = decode_loop.build_decode_state(cfg, bs, ctx, ps, args.e2e_steps + 5 * 8 + 40, use_graph=False)
        samples = 5
        each = max(1, args.e2e_steps // samples)
        dts = decode_loop.time_decode_samples(mr, batch, samples, each, 4, barrier=barrier)
        if world > 1:
            from mini_sglang_amd.comm import agree

            dts = [agree(dt, "max", "cuda") for dt in dts]
        sp = spread([dt / each * 1e3 for dt in dts], 3)
        info.update({"tok_per_s": round(bs / (sp["median"] * 1e-3), 1), "ms_per_step": sp["median"], "ms_per_step_spread": sp,
                     "steps": samples * each, "timing": f"median of {samples} samples of {each} steps", "hip_graph": use_graph})
        if use_graph:
            try:  # the same steps chained on the device, 8 per host synchronisation
                dts = decode_loop.time_decode_chained_samples(mr, batch, samples, 8, 2, barrier=barrier)
                if world > 1:
                    dts = [agree(dt, "max", "cuda") for dt in dts]
                sp = spread([dt / 8 * 1e3 for dt in dts], 3)
                info.update({"tok_per_s_chained": round(bs / (sp["median"] * 1e-3), 1), "ms_per_step_chained": sp["median"],
                             "ms_per_step_chained_spread": sp, "chain": 8})
            except Exception as e:
                info["chained_error"] = f"{type(e).__name__}: {e}"[:200]
        return info
    except Exception as e:  # the headline number above must survive a failure here
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def allreduce_trial_child(world, rank, local_rank):
    """One rank of the direct-read all-reduce trial in a process of its own (see tp_allreduce_trial_isolated): a gloo
    group for the set-up exchange, this rank's GPU, the trial, one line of JSON."""
    import torch.distributed as dist

    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    info = tp_allreduce_trial(world, rank, dist.barrier)
    print("TRIAL " + json.dumps(info), flush=True)
    dist.destroy_process_group()


def tp_allreduce_trial_isolated(world, rank, local_rank, barrier):
    """The trial below, run by a child process per rank: a kernel that reads peer memory wrongly on this node's
    links would take its process down, and that must not be the process that owes the benchmark line.  Only
    when every child reports bit-exact results does the parent use the communicator itself (timing beside RCCL,
    second decode measurement)."""
    import subprocess

    import torch.distributed as dist

    info = {"validated": False}
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}  # the children rendezvous among themselves
    env.update(MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 17), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    child = None
    try:
        child = subprocess.run([sys.executable, os.path.abspath(__file__), "--allreduce-trial-child"], env=env, capture_output=True,
                               text=True, timeout=150)
        lines = [ln for ln in child.stdout.splitlines() if ln.startswith("TRIAL ")]
        if child.returncode == 0 and lines:
            info = json.loads(lines[-1][6:])
        else:
            info["error"] = f"child exit {child.returncode}: {(child.stderr or child.stdout)[-300:]}"
    except Exception as e:  # noqa: BLE001 (timeout included: subprocess.run has killed the child)
        info["error"] = f"{type(e).__name__}: {e}"[:300]
    dev = torch.device("cuda", torch.cuda.current_device())
    from mini_sglang_amd.comm import agree

    info["validated"] = bool(agree(1.0 if info.get("validated") else 0.0, "min", dev))
    info["isolated"] = True
    if info["validated"] and "sizes" in info and dist.get_backend() == "nccl":  # RCCL on the same sizes, in this process
        for name, entry in info["sizes"].items():
            x = torch.zeros(int(name.split("x")[0]) * 4096, dtype=torch.bfloat16, device=dev)
            for _ in range(10):
                dist.all_reduce(x)
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                dist.all_reduce(x)
            e1.record()
            torch.cuda.synchronize()
            entry["rccl_us"] = round(agree(e0.elapsed_time(e1) * 10.0, "max", dev), 2)
    return info


def tp_allreduce_trial(world, rank, barrier):
    """The direct-read all-reduce (include/msgl_comm.h) next to RCCL's on the sizes a decode step reduces
    ([bs, 4096] bf16 for bs 1 .. 256).  It is only used for the second decode_e2e measurement when every rank's
    result equals the fp32 rank-order sum of the gathered inputs bit for bit and no kernel timed out."""
    import torch.distributed as dist

    from mini_sglang_amd.comm import XgmiAllReduce, agree, tp_all_gather

    info = {"validated": False}
    try:
        comm = XgmiAllReduce(max_bytes=8 << 20)  # raises on every rank or on none
        # ranks that share a device (a one-GPU rehearsal) get every call issued step by step: results are checked
        # all the same, the times then say nothing about a node with a GPU per rank
        info["stepped"] = bool(comm.stepped)
        dev = torch.device("cuda", torch.cuda.current_device())
        cases = []
        for n in (4096, 64 * 4096, 256 * 4096):  # torch collectives first ...
            g = torch.Generator(device=dev).manual_seed(17 * n + rank)
            x = torch.randn(n, device=dev, generator=g).bfloat16()
            parts = tp_all_gather(x)
            want = torch.zeros(n, dtype=torch.float32, device=dev)
            for p in parts:
                want += p.float()
            cases.append((x, want.bfloat16()))
        ok = True
        try:  # ... then only the kernels under test, so that a local failure cannot desynchronise the ranks
            for x, want in cases:
                got = comm.all_reduce(x.clone())
                comm.check()
                ok = ok and torch.equal(got, want)
        except Exception as e:  # noqa: BLE001
            ok = False
            info["error"] = f"{type(e).__name__}: {e}"[:300]
        info["validated"] = bool(agree(1.0 if ok else 0.0, "min", dev))
        if not info["validated"]:
            return info

        def timed(fn, x, iters=100):
            for _ in range(10):
                fn(x)
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn(x)
            e1.record()
            torch.cuda.synchronize()
            return round(agree(e0.elapsed_time(e1) / iters * 1e3, "max", dev), 2)

        sizes = {}
        for bs in (1, 16, 64, 256):
            x = torch.zeros(bs * 4096, dtype=torch.bfloat16, device=dev)
            sizes[f"{bs}x4096"] = {"xgmi_us": timed(comm.all_reduce, x)}
            if dist.get_backend() == "nccl":  # (the CPU-side gloo group of the tests has nothing to compare with)
                sizes[f"{bs}x4096"]["rccl_us"] = timed(dist.all_reduce, x)
        comm.check()
        info["sizes"] = sizes
        comm.close()
    except Exception as e:
        info["validated"] = False
        info["error"] = f"{type(e).__name__}: {e}"[:300]
    return info


def cpu_baseline(pool, q, r2t, S, ps, Hq, Hkv, D, scale, budget_s):
    """The oracle (the reference's per-request gather + SDPA loop, torch_attn_backend.py:114-182) timed on
    the host cores over a bounded sample of the same workload: the first n requests of layer 0, with their
    K/V rows copied to host memory at scattered slots."""
    from oracle import torch_oracle as orc

    # threads: the CPUs this process may run on (its affinity mask), further limited by OMP_NUM_THREADS when the box
    # sets one (a GPU box hands a job a share of the machine, not all of os.cpu_count()); both numbers are reported
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = max(1, min(allowed, omp) if omp > 0 else allowed)
    torch.set_num_threads(cores)
    kc, vc = pool

    def sample(n):
        slots_dev = r2t[:n, :S].reshape(-1).long()
        k_h = kc[slots_dev].cpu()
        v_h = vc[slots_dev].cpu()
        # keep the gather scattered on the host too: place row i at a pseudo-random slot
        from mini_sglang_amd.runtime import synth
        perm = synth.permutation(n * S, 11)
        k_pool = torch.empty_like(k_h)
        v_pool = torch.empty_like(v_h)
        k_pool[perm] = k_h
        v_pool[perm] = v_h
        table = perm.to(torch.int32).view(n, S)
        return k_pool, v_pool, table

    n = 2
    k_pool, v_pool, table = sample(n)
    qh = q[:n].cpu()
    idx = torch.arange(n, dtype=torch.int32)
    lens = torch.full((n,), S, dtype=torch.int32)
    orc.paged_decode_ref(qh, k_pool, v_pool, table, idx, lens, scale)  # warm-up
    t0 = time.perf_counter()
    orc.paged_decode_ref(qh, k_pool, v_pool, table, idx, lens, scale)
    per_req = (time.perf_counter() - t0) / n
    n = int(max(2, min(q.shape[0], budget_s / max(per_req, 1e-6))))
    k_pool, v_pool, table = sample(n)
    qh = q[:n].cpu()
    idx = torch.arange(n, dtype=torch.int32)
    lens = torch.full((n,), S, dtype=torch.int32)
    reps = 0
    t0 = time.perf_counter()
    while True:
        orc.paged_decode_ref(qh, k_pool, v_pool, table, idx, lens, scale)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or reps >= 64:
            break
    gbps = 2 * Hkv * D * 2 * S * n * reps / dt / 1e9
    return {"value": round(gbps, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"first {n} of the batch's {q.shape[0]} requests ({S} tokens each), one layer, {reps} passes, {dt:.1f} s of host time",
            "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(), "cpus_allowed": allowed,
            "threads_note": "threads = min(CPUs in this process's affinity mask, OMP_NUM_THREADS if the box sets it)"}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
