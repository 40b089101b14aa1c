"""-m gpu: the N > 1 flow of bench.py with REAL HIP renders per rank: two ranks under torch.distributed.run share the
one GPU of the box (gloo, bands travel via host memory), each renders its interleaved bands of ONE frame through the C
ABI, rank 0 gathers and de-interleaves them and checks 4 000 pixels of the assembled frame against the CPU oracle.
(Over RCCL / xGMI the same flow is the driver's multi-GPU run; here every step except the transport is the product's.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("workload,ranks", [("cfg2", 2), ("cfg3", 3)])        # cfg3: 1080 rows over 3 ranks, ragged bands
def test_bench_ranks_share_one_gpu(workload, ranks):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--workload", workload, "--backend", "gloo"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == ranks and out["scaling"] == "strong"
    assert out["gathered_frame_matches_oracle_on_sample"] is True
    # the same frame as one rank renders it
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--workload", workload,
                          "--timed-only"], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert out["frame_checksum"] == ref["frame_checksum"]


def _bench(*args):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout must be the one JSON line: %r" % lines[:3]
    return json.loads(lines[0])


def test_bench_collective_flow_over_rccl_on_one_rank():
    """The N > 1 step of bench.py over the REAL collective library (backend nccl = RCCL) with the one rank a one-GPU box has:
    process group, two contexts on two streams, gather, de-interleave — the frame must be the one the plain N = 1 run renders,
    and stdout the one JSON line (RCCL's banner goes to stderr)."""
    one = _bench("--steps", "1", "--warmup", "0", "--workload", "cfg2", "--timed-only")
    coll = _bench("--steps", "3", "--warmup", "1", "--workload", "cfg2", "--force-collective", "--no-cpu-baseline")
    assert coll["n_gpus"] == 1 and coll["frame_checksum"] == one["frame_checksum"]
    assert coll["ms_per_step"] > 0 and coll["roofline"]["bound"] == "valu"


def test_bench_emulated_rank_of_four():
    """--emulate-rank r/N: one process plays rank r of an N-rank job and the root at once (its bands through two contexts, the
    RCCL gather into slot r, the root's de-interleave over N slots); the bands it rendered are checked against the oracle."""
    out = _bench("--steps", "3", "--warmup", "1", "--workload", "cfg2", "--emulate-rank", "1/4")
    assert "emulated_rank" in out
    assert out["gathered_frame_matches_oracle_on_sample"] is True
    assert out["host_enqueue_ms_per_step"] >= 0.0
