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
