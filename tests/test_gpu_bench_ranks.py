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
