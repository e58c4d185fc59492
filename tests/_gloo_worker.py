"""Worker of tests/test_multigpu_gloo.py: one rank of a world_size-2 (or -8) gloo group.

argv: out_dir [golden case = case2_two_samples] [samples = the case's own].  With more samples than the case holds, sample i is
the case's column i % S under the name <name>_<i> (the 8-rank test: case 5's three samples dealt to eight ranks).
Each rank owns one sample.  The per-sample tables are computed from the
reference's mapped.csv (numpy stands in for the kernels: the point of this test is the
sharding / gather / merge logic of multigpu.py, which has no GPU in it), gathered on rank 0 and
written with finish_tables; the parent compares the CSVs with the reference's."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import GoldenCase  # noqa: E402
import mirge3_amd  # noqa: E402,F401
from mirge3_amd import multigpu  # noqa: E402
from mirge3_amd.countjoin import finish_tables  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # the run's directory: without -onam its name carries the start time to the second; the ranks of this test reach this line
    # more than a second apart and must still end up with rank 0's choice (multigpu.agree_on_run_directory)
    import time
    time.sleep(1.2 * rank)
    name = multigpu.agree_on_run_directory(None, rank, world, dist)
    names = [None] * world
    dist.all_gather_object(names, name)
    assert names == [names[0]] * world and name.startswith("miRge."), names
    assert multigpu.agree_on_run_directory("given", rank, world, dist) == "given"
    # a directory every rank sees rank 0's files in -> hand-over through files; one that a rank does not see them in (here: a
    # different path per rank stands in for another node's local disk) -> that rank sends in-band
    assert multigpu.directory_is_shared(out_dir, rank, world, dist) is True
    private = os.path.join(out_dir, f"node{rank}")
    os.makedirs(private, exist_ok=True)
    assert multigpu.directory_is_shared(private, rank, world, dist) is (rank == 0)
    case = GoldenCase(sys.argv[2] if len(sys.argv) > 2 else "case2_two_samples")
    n_samples = int(sys.argv[3]) if len(sys.argv) > 3 else len(case.samples)
    col_of = [i % len(case.samples) for i in range(n_samples)]
    name_of = [case.samples[c] if n_samples == len(case.samples) else f"{case.samples[c]}_{i}" for i, c in enumerate(col_of)]
    exp = case.expected_annotation()
    mir = case.libs["mirna"]
    lut = {}
    for i, nm in enumerate(mir.names):
        lut.setdefault(nm, i)

    def process(i):
        cls = np.zeros(case.n_pass, dtype=np.int64)
        ex = np.zeros(len(mir), dtype=np.int64)
        iso = np.zeros(len(mir), dtype=np.int64)
        ci = col_of[i]
        for s, row in zip(case.seqs, case.counts):
            p, nm = exp[s]
            c = int(row[ci])
            if p < 0 or c == 0:
                continue
            cls[p] += c
            if p == 0:
                ex[lut[nm]] += c
            if p == 8:
                iso[lut[nm]] += c
        nme, src = name_of[i], case.samples[ci]
        t = multigpu.SampleTables(i, nme, case.sample_read_counts[src], case.trimmed[src],
                                  case.trimmed_unique[src], cls, ex, iso)
        # the sample's dictionary travels with its tables (what rank 0 merges into the run's one mapped.csv)
        from mirge3_amd.seqio import FlatSeqs
        mine = [(s, int(row[ci])) for s, row in zip(case.seqs, case.counts) if int(row[ci])]
        fs = FlatSeqs.from_list([s for s, _ in mine])
        t.reads = multigpu.SampleReads.from_seqs(fs, np.array([c for _, c in mine], dtype=np.uint32))
        if i % 2:  # the hand-over through files in the run's directory (what fastpath.run_sample_tables does), and in-band
            t.reads = t.reads.to_files(os.path.join(out_dir, ".mirge_shards"), i)
        return t

    assert multigpu.assign_samples(n_samples, world)[rank] == list(range(rank, n_samples, world))
    tables = multigpu.run_sharded(n_samples, rank, world, process, dist)
    if rank == 0:
        for t in tables:  # every rank's dictionary arrived whole, in sample order
            if isinstance(t.reads, str):
                t.reads = multigpu.SampleReads.from_files(t.reads)
            src = case.samples[col_of[t.index]]
            assert t.name == name_of[t.index]
            assert int(t.reads.counts.sum()) == case.trimmed[src] and len(t.reads.counts) == case.trimmed_unique[src]
            assert t.reads.offsets.shape[0] == len(t.reads.counts) + 1 and t.reads.data.shape[0] == int(t.reads.offsets[-1])
        names, src, trimmed, uniq, cls, ex, iso = multigpu.merge_tables(tables)
        assert [t.index for t in tables] == list(range(n_samples)) and names == name_of
        finish_tables(cls, ex, iso, mir, case.merges, names, src, trimmed, uniq, float(case.cr), case.spike, workDir=out_dir)
    else:
        assert tables is None
    dist.barrier()
    range_tail(out_dir, rank, world, case, n_samples, col_of, name_of, exp)
    dist.barrier()
    dist.destroy_process_group()


def range_tail(out_dir, rank, world, case, n_samples, col_of, name_of, exp):
    """The parallel tail's host protocol (multigpu.py 'The parallel tail', fastpath.run_sharded_ranges) with numpy / Python standing in
    for the four device steps (quantile keys, range split, merge + order, row formatting): pooled quantiles -> the same splitters on
    every rank -> every dictionary cut by owner range and handed over as files -> each rank merges ITS range of every sample, sorts it,
    formats its rows, the ranks' byte counts give the offsets, rank 0 creates the files, every rank pwrites its stretch.  The parent
    compares mapped.csv / unmapped.csv with the reference's."""
    from helpers import PASS_COLS, lex_key0
    from mirge3_amd.seqio import FlatSeqs
    shard = os.path.join(out_dir, ".mirge_range_shards")
    k = 16
    mine = {}
    for i in multigpu.assign_samples(n_samples, world)[rank]:
        d = [(s, int(row[col_of[i]])) for s, row in zip(case.seqs, case.counts) if int(row[col_of[i]])]
        keys = np.array([lex_key0(s) for s, _ in d], dtype=np.uint64)
        srt = np.sort(keys)
        qs = srt[np.minimum(len(srt) - 1, (2 * np.arange(k) + 1) * len(srt) // (2 * k))] if len(srt) else np.full(k, multigpu.KEY_NONE)
        mine[i] = (d, keys, qs)
    pool = [None] * world
    dist.all_gather_object(pool, [(i, len(v[0]), v[2]) for i, v in sorted(mine.items())])
    flat = sorted((x for part in pool for x in part), key=lambda x: x[0])
    assert [x[0] for x in flat] == list(range(n_samples))
    sp = multigpu.choose_splitters([(u, q) for _, u, q in flat], world)
    every_sp = [None] * world
    dist.all_gather_object(every_sp, sp.tolist())
    assert every_sp == [every_sp[0]] * world and len(sp) == world - 1  # the same cut everywhere, without a broadcast
    for i, (d, keys, _) in mine.items():
        owner = np.searchsorted(sp, keys, side="right")
        o = np.argsort(owner, kind="stable")
        bounds = np.concatenate(([0], np.cumsum(np.bincount(owner, minlength=world)))).astype(np.int64)
        multigpu.write_parts(shard, i, FlatSeqs.from_list([d[j][0] for j in o]), np.array([d[j][1] for j in o], dtype=np.uint32), bounds)
    dist.barrier()
    joint = {}
    for i in range(n_samples):
        part = multigpu.read_part(shard, i, rank)
        for q, c in zip(FlatSeqs(part.data, part.offsets).to_list(), part.counts):
            joint.setdefault(q, [0] * n_samples)[i] = int(c)
    n_cols = case.n_pass
    rows_m, rows_u = [], []
    for q in sorted(joint):
        p, nm = exp[q]
        cols = [""] * n_cols
        if p >= 0:
            cols[p] = nm
        line = ",".join([q, "1" if p >= 0 else "0"] + cols + [str(c) for c in joint[q]]) + "\n"
        (rows_m if p >= 0 else rows_u).append(line)
    text_m, text_u = "".join(rows_m).encode(), "".join(rows_u).encode()
    every = [None] * world
    dist.all_gather_object(every, (len(text_m), len(text_u), max(joint) if joint else None, min(joint) if joint else None))
    header = (",".join(["Sequence", "annotFlag"] + PASS_COLS[:n_cols] + name_of) + "\n").encode()
    paths = [os.path.join(out_dir, f) for f in ("mapped.csv", "unmapped.csv")]
    if rank == 0:
        for w, path in enumerate(paths):
            with open(path, "wb") as fh:
                fh.write(header)
                fh.truncate(len(header) + sum(x[w] for x in every))
        # the ranges are consecutive stretches of the sorted union: a rank's largest read sorts before the next rank's smallest
        held = [x for x in every if x[2] is not None]
        assert all(a[2] < b[3] for a, b in zip(held, held[1:]))
    dist.barrier()
    for w, (path, text) in enumerate(zip(paths, (text_m, text_u))):
        fd = os.open(path, os.O_WRONLY)
        os.pwrite(fd, text, len(header) + sum(x[w] for x in every[:rank]))
        os.close(fd)
    dist.barrier()
    if rank == 0:
        assert not os.listdir(shard)
        os.rmdir(shard)


if __name__ == "__main__":
    main()
