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
    # round 6: the run's ONE mapped.csv / unmapped.csv written range by range by BOTH ranks (the worker's range_tail)
    for f in ("mapped.csv", "unmapped.csv"):
        assert (tmp_path / f).read_text() == case.text(f), f


def test_eight_ranks_gloo_eight_dictionaries(tmp_path):
    """The N = 8 path without hardware (BASELINE configs[3]: 8 samples, one per GPU): eight gloo ranks, `assign_samples(8, 8)`,
    golden case 5's three samples (spike-in library) dealt to eight samples, eight dictionaries handed to rank 0 (files and
    in-band), eight columns merged.  Every column of the run's tables must be the reference's column of the sample it repeats."""
    import csv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(HERE, "_gloo_worker.py"), str(tmp_path),
           "case5_three_samples_spikein", "8"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    case = GoldenCase("case5_three_samples_spikein")
    S = len(case.samples)
    for f in ("miR.Counts.csv", "miR.RPM.csv"):
        got = list(csv.reader((tmp_path / f).read_text().splitlines()))
        want = list(csv.reader(case.text(f).splitlines()))
        assert got[0] == [want[0][0]] + [f"{want[0][1 + i % S]}_{i}" for i in range(8)]
        assert [g[0] for g in got] == [w[0] for w in want]
        for g, w in zip(got[1:], want[1:]):
            assert g[1:] == [w[1 + i % S] for i in range(8)], (f, g[0])
    # round 6: the range hand-over at eight ranks -- every rank wrote its stretch of the two per-read files: the reference's rows,
    # in the reference's (sorted) order, the case's three count columns dealt to the eight samples
    for f in ("mapped.csv", "unmapped.csv"):
        got = list(csv.reader((tmp_path / f).read_text().splitlines()))
        want = list(csv.reader(case.text(f).splitlines()))
        n_fixed = len(want[0]) - S
        assert got[0] == want[0][:n_fixed] + [f"{want[0][n_fixed + i % S]}_{i}" for i in range(8)]
        assert len(got) == len(want)
        for g, w in zip(got[1:], want[1:]):
            assert g == w[:n_fixed] + [w[n_fixed + i % S] for i in range(8)], (f, g[0])
    got = list(csv.reader((tmp_path / "annotation.report.csv").read_text().splitlines()))
    want = list(csv.reader(case.text("annotation.report.csv").splitlines()))
    assert got[0] == want[0] and len(got) == 9
    for i, g in enumerate(got[1:]):
        assert g[0] == f"{want[1 + i % S][0]}_{i}" and g[1:] == want[1 + i % S][1:]
