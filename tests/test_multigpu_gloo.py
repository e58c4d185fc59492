"""N > 1 path on CPU: world_size 2, gloo, 127.0.0.1 (no GPU involved)."""
import os
import subprocess
import sys

from helpers import GoldenCase
import mirge3_amd  # noqa: F401
from mirge3_amd import multigpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_assign_samples():
    assert multigpu.assign_samples(8, 8) == [[i] for i in range(8)]
    assert multigpu.assign_samples(5, 2) == [[0, 2, 4], [1, 3]]
    assert multigpu.assign_samples(1, 4) == [[0], [], [], []]


def test_two_ranks_gloo_merge_equals_reference(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(HERE, "_gloo_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    case = GoldenCase("case2_two_samples")
    for f in ("annotation.report.csv", "annotation.report.html", "miR.Counts.csv", "miR.RPM.csv"):
        assert (tmp_path / f).read_text() == case.text(f), f
