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


def test_choose_splitters_and_part_files(tmp_path):
    """multigpu.choose_splitters / write_parts / read_part (round 6, the parallel tail's host side; no GPU): weighted quantiles of the
    pooled keys -- a sample's k keys each stand for U / k of its reads --, ascending, the same whatever order the pool is handed
    over in; empty dictionaries vote for nothing; a dictionary cut by bounds comes back stretch by stretch."""
    import numpy as np
    from mirge3_amd.seqio import FlatSeqs
    rng = np.random.default_rng(3)
    big = np.sort(rng.integers(0, 1 << 60, size=512).astype(np.uint64))
    small = np.sort(rng.integers(0, 1 << 40, size=512).astype(np.uint64))      # a sample whose reads all sort low
    empty = np.full(512, multigpu.KEY_NONE, dtype=np.uint64)
    sp = multigpu.choose_splitters([(1_000_000, big), (1_000, small), (0, empty)], 8)
    assert sp.shape == (7,) and sp.dtype == np.uint64 and (np.diff(sp.astype(np.float64)) >= 0).all()
    assert np.array_equal(sp, multigpu.choose_splitters([(0, empty), (1_000, small), (1_000_000, big)], 8))
    # the big sample carries 1000 x the weight: the cuts are (nearly) its own octiles, the small sample's keys hardly move them
    assert np.abs(np.searchsorted(big, sp) - 64 * np.arange(1, 8)).max() <= 2
    assert multigpu.choose_splitters([(5, big)], 1).shape == (0,)
    assert (multigpu.choose_splitters([(0, empty)], 4) == np.uint64(1) << np.uint64(63)).all()
    seqs = ["ACGT" * 5, "T" * 17, "GGGCCCAAATTTGGGC", "A" * 300, "CCCCCCCCCCCCCCCCCC"]
    fs = FlatSeqs.from_list(seqs)
    cnt = np.array([[3], [1], [7], [2], [9]], dtype=np.uint32)
    multigpu.write_parts(tmp_path, 4, fs, cnt, np.array([0, 2, 2, 5], dtype=np.int64))
    got = [multigpu.read_part(tmp_path, 4, q) for q in range(3)]
    assert [FlatSeqs(p.data, p.offsets).to_list() for p in got] == [seqs[:2], [], seqs[2:]]
    assert [p.counts.tolist() for p in got] == [[3, 1], [], [7, 2, 9]] and got[2].lengths.dtype == np.uint16
    assert not list(tmp_path.iterdir())  # read_part removes what it has read


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
