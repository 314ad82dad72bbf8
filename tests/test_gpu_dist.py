"""N>1 path on the GPU box: the sharded plans + exchange (tpch_dist.py) must reproduce the
single-GPU results bit-exactly.  The box has ONE GPU, so the two ranks share it and the
collectives run over gloo (host-staged); on a multi-GPU node the same script runs over RCCL."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_plans_match_single_gpu(world):
    env = dict(os.environ, LDB_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_check.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("OK") == 13, r.stdout  # 12 queries + the NULL exchange
