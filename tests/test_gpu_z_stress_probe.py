"""tools/stress_probe.py under pytest: every probe kind on the run-time specialised + lazily fused path (12 M probe
rows: above both thresholds) against numpy, WITH torch initialised in the process — torch binds its own older
libhiprtc / libamdhip64, the configuration bench.py runs in and the one two shapes of the specialised tile
kernels once misbehaved under (DESIGN §2 "hiprtc under torch").  debug_check range-checks every produced row id.
Both table layouts: direct-addressed (the default for primary keys) and open addressing (join_direct = 0)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("direct", [1, 0])
def test_probe_kinds_against_numpy_with_torch_loaded(direct):
    env = dict(os.environ, LDB_DEBUG_CHECK="1", LDB_JOIN_DIRECT=str(direct))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_probe.py"), "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout and "MISMATCH" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
