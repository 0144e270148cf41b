"""Direct-read all-reduce (include/msgl_comm.h) with 2 .. 8 ranks, one process each, against the fp32 rank-order
sum it is specified to return (bit-exact), in both forms and back to back with changing grid sizes.

The test box has one GPU, so the ranks share it.  Whether kernels of different processes then run at the same
time is the driver's business, and a kernel that waits for its peer's kernel may wait in vain; the communicator
therefore issues the calls of ranks that share a device step by step (same device code, host barriers at the
protocol's synchronisation points), and those tests are deterministic.  The one-launch form, the one a node with
a GPU per rank runs, is tested after a probe has shown that the two processes' kernels do overlap here, and is
skipped with that reason otherwise."""
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_tp2_model_on_direct_read_allreduce_matches_reference_stream():
    """The whole stack, tensor parallel over two ranks (KV heads, MLP columns and vocabulary sharded), every
    all-reduce of hidden states on the direct-read kernels, must generate the single-rank greedy stream of the
    reference fixture, on both ranks."""
    import json

    import golden_cases as gc

    outs = _run_ranks(2, "tp-model")
    fx = gc.load("e2e_tiny_ps16.pt")
    for out in outs:
        line = [ln for ln in out.splitlines() if ln.startswith("TOKENS ")][-1]
        assert json.loads(line[7:]) == fx["tokens"][:, :24].tolist()


def _run_ranks(world, *args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "ar_worker.py"), *args], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs, failed = [], False
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            failed = True
            for q in procs:
                if q.poll() is None:
                    q.kill()  # exactly the processes started above
            out, _ = p.communicate()
            out += "\n[timed out]"
        outs.append(out)
    report = "\n".join(f"--- rank {r} (exit {p.returncode})\n{o[-2000:]}" for r, (p, o) in enumerate(zip(procs, outs)))
    print(report)
    assert not failed and all(p.returncode == 0 for p in procs), report
    return outs


@pytest.mark.parametrize("world", [2, 3, 4, 6, 8])  # 3 and 6 take the kernels without a compile-time rank count
def test_allreduce_matches_rank_order_fp32_sum(world):
    outs = _run_ranks(world, "stepped")
    assert all("(stepped): OK" in o for o in outs)


@pytest.mark.parametrize("world", [2, 4])
def test_one_launch_form_when_the_ranks_kernels_overlap(world):
    """Every size in both forms, 36 back-to-back calls whose grid size and form change from call to call with a
    lagging rank, and hipGraph replay, all in the one-launch form."""
    outs = _run_ranks(world, "fused")
    if any("NOT-CONCURRENT" in o for o in outs):
        pytest.skip("kernels of two processes do not run side by side on this GPU; the one-launch form needs a GPU per rank")
    assert all("(fused): OK" in o for o in outs)


def test_bench_trial_validates_and_times():
    """The trial bench.py runs, one child process per rank, before it trusts the direct-read all-reduce for the
    decode measurement: here the children themselves, two ranks on this box's GPU."""
    import json

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="2")
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    procs = [subprocess.Popen([sys.executable, "-u", bench, "--allreduce-trial-child"], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for p in procs:
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            raise
        lines = [ln for ln in out.splitlines() if ln.startswith("TRIAL ")]
        assert p.returncode == 0 and lines, out[-2000:]
        info = json.loads(lines[-1][6:])
        assert info["validated"] and set(info["sizes"]) == {"1x4096", "16x4096", "64x4096", "256x4096"}, info


def test_bench_tensor_parallel_control_flow_end_to_end(tmp_path):
    """bench.py's N > 1 path from start to finish, two ranks on this box's GPU over gloo: headline measurement with
    heads divided over the ranks, the secondary shapes, the all-reduce trial in child processes, and both whole-model
    decode legs (eager, hidden-state all-reduces on the direct-read kernels) — so that the first run on a real
    multi-GPU node is not also the first execution of that code.  Only the plumbing is checked, not the numbers."""
    import json

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="2")
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    cmd = [sys.executable, "-u", bench, "--gpus", "2", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1", "--bs", "16", "--ctx", "1024",
           "--layers", "2", "--e2e-steps", "2", "--e2e-ctx", "512", "--mixed-bs", "16", "--secondary-deadline", "400"]
    env["MSGL_BENCH_TRACE"] = "1"  # a progress line per leg on stderr: a rank that stops says where
    logs = [open(os.path.join(str(tmp_path), f"rank{r}.log"), "w+") for r in range(2)]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=logs[r], stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    deadline = time.monotonic() + 150
    while any(p.poll() is None for p in procs) and time.monotonic() < deadline:
        time.sleep(1)
    late = [p for p in procs if p.poll() is None]
    for p in late:
        p.kill()
        p.wait()
    outs = []
    for f in logs:
        f.seek(0)
        outs.append(f.read())
        f.close()
    assert not late and all(p.returncode == 0 for p in procs), "\n".join(f"--- rank {r}\n{o[-3000:]}" for r, o in enumerate(outs))
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert lines, outs[0][-3000:]
    res = json.loads(lines[-1])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["config"]["hkv_per_gpu"] == 4 and res["scaling"] == "strong"
    assert "secondary_timeout_s" not in res
    assert res["tp_allreduce"]["validated"], res["tp_allreduce"]
    for leg in ("decode_e2e", "decode_e2e_mixed"):
        assert "error" not in res[leg] and "skipped" not in res[leg] and res[leg]["tok_per_s"] > 0, (leg, res[leg])
        assert res[leg]["tp"] == 2 and res[leg]["hip_graph"] is False
    assert "error" not in res["other_configs"] and "error" not in res["extend"], (res["other_configs"], res["extend"])
    assert "error" not in res["layers"] and res["layers"]["ops"], res["layers"]
    # what lets a reader of a multi-GPU record check "N ranks on N distinct GPUs": here two ranks on ONE device
    d = res["distributed"]
    assert d["world_size"] == 2 and d["backend"] == "gloo" and "rccl_version" in d
    assert len(d["devices"]) == 2 and d["devices"][0] == d["devices"][1] and d["distinct_devices"] == 1
    assert len(d["rank_launch_ms"]) == 2 and all(ms > 0 for ms in d["rank_launch_ms"])
