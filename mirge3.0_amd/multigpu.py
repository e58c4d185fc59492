"""Sample sharding over the GPUs of one node.

The reference processes samples one after the other in one process (``digest.py:133``) and only
joins their columns at the end (``digest.py:243``; every later statistic is per sample column).  So
the path shards by sample with **no exchange step**: one process per GPU, each rank runs
collapse -> cascade -> count join on its own samples against its own copy of the libraries, and rank 0
gathers the per-sample tables (a few kB each) to write the run's CSVs -- and, for the run's one ``mapped.csv`` /
``unmapped.csv``, each sample's dictionary with its annotation (``SampleReads``, ~15 B per unique read), which it
merges into the sample matrix with one weighted collapse on its GPU (``fastpath.merge_sample_reads``).  The gather is
result collection on the host side (``torch.distributed.gather_object``), not a data-path collective.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


def assign_samples(n_samples: int, world: int) -> List[List[int]]:
    """Round-robin: sample i goes to rank i % world (one sample per GPU when n_samples == world)."""
    out: List[List[int]] = [[] for _ in range(world)]
    for i in range(n_samples):
        out[i % world].append(i)
    return out


@dataclass
class SampleReads:
    """A sample's dictionary as rank 0 needs it for the run's joint per-read tables (``mapped.csv`` / ``unmapped.csv``,
    mirge/__main__.py:164-173; -gff / -ai / -ie): unique reads in order of first appearance and their counts.  (Until round 5
    the annotation travelled too; rank 0 now annotates the joint table itself.)"""
    data: np.ndarray     # uint8, the sequences' ASCII
    lengths: np.ndarray  # uint8 / uint16 / int64 [U]: the narrowest type that holds the longest read
    counts: np.ndarray   # uint32 [U]
    iupac: bool = False

    FIELDS = ("data", "lengths", "counts")

    @staticmethod
    def from_seqs(seqs, counts, iupac: bool = False) -> "SampleReads":
        ln = seqs.lengths
        mx = int(ln.max()) if len(ln) else 0
        ln = ln.astype(np.uint8 if mx < 256 else np.uint16 if mx < 65536 else np.int64)
        return SampleReads(np.ascontiguousarray(seqs.data), ln, np.ascontiguousarray(counts, dtype=np.uint32), iupac)

    @property
    def offsets(self) -> np.ndarray:
        off = np.zeros(self.lengths.shape[0] + 1, dtype=np.int64)
        np.cumsum(self.lengths, out=off[1:], dtype=np.int64)
        return off

    def to_files(self, directory, index: int) -> str:
        """the arrays as plain .npy files (a rank's hand-over to rank 0 on the same node: 30 B per unique read would
        otherwise be pickled and pushed through the gather's TCP loopback); returns the stem to ``from_files``"""
        import os
        os.makedirs(directory, exist_ok=True)
        stem = os.path.join(str(directory), f"sample{index}")
        for f in self.FIELDS:
            np.save(stem + "." + f + ".npy", np.ascontiguousarray(getattr(self, f)))
        return stem + ("|1" if self.iupac else "|0")

    @staticmethod
    def from_files(ref: str, remove: bool = True) -> "SampleReads":
        import os
        stem, iupac = ref.rsplit("|", 1)
        arr = [np.load(stem + "." + f + ".npy") for f in SampleReads.FIELDS]
        if remove:
            for f in SampleReads.FIELDS:
                os.unlink(stem + "." + f + ".npy")
        return SampleReads(*arr, iupac == "1")


@dataclass
class SampleTables:
    """What one sample contributes to the run's tables (all int64, S = 1 column)."""
    index: int
    name: str
    total_input: int
    trimmed_all: int
    trimmed_unique: int
    class_sums: np.ndarray  # [n_pass]
    exact: np.ndarray       # [n_mirna]
    iso: np.ndarray         # [n_mirna]
    reads: Optional[object] = None  # SampleReads, or the stem of its files (SampleReads.to_files)
    timing: Optional[dict] = None   # the rank's own stage times for this sample (run.log of the sharded run)
    iupac: bool = False             # the sample held IUPAC codes other than N (run.log says so)


# ---------------------------------------------------------------------------------------------------------------------
# The parallel tail (round 6).  The reference puts ONE mapped.csv / unmapped.csv over the sorted union of all samples'
# sequences, one count column per sample (digest.py:243, mirge/__main__.py:164-173).  Until round 5 rank 0 built that table
# alone while the other GPUs idled (1.31 s at 8 x 20 M reads against ~0.3 s for a rank's own sample).  Now the union's key
# space is cut into one RANGE per rank: a read's owner follows from the first 21 bases of its sort key, ranges are
# consecutive stretches of the sorted union, so rank q merges the samples' reads of range q (weighted collapse, S columns),
# annotates them (one cascade), orders them (device sort) and formats + pwrites its stretch of the two files at the byte
# offset the ranks' sizes give.  Hand-over: files in the run's directory (or /dev/shm), 30 B per unique read; no RCCL.
# ---------------------------------------------------------------------------------------------------------------------
RANGE_SAMPLE_KEYS = 512   # quantiles a sample contributes to the splitter pool (a range is then balanced to ~1/512 of a sample)
KEY_NONE = np.uint64(0xFFFFFFFFFFFFFFFF)  # what an empty dictionary's quantiles read (mirge_reads_range_sample)


def choose_splitters(pool, n_parts: int) -> np.ndarray:
    """``pool`` = [(unique reads of a sample, its k quantile keys)] over ALL samples of the run -> the n_parts - 1 ascending
    splitter keys (uint64): weighted quantiles of the pooled keys, a sample's k keys each standing for U / k of its reads.
    Every rank computes this from the same all-gathered pool, so all agree without a broadcast."""
    keys, wts = [], []
    for u, k in pool:
        k = np.asarray(k, dtype=np.uint64)
        k = k[k != KEY_NONE]
        if u and k.size:
            keys.append(k)
            wts.append(np.full(k.shape[0], float(u) / k.shape[0]))
    if n_parts <= 1:
        return np.zeros(0, dtype=np.uint64)
    if not keys:
        return np.full(n_parts - 1, np.uint64(1) << np.uint64(63), dtype=np.uint64)  # beyond every key: range 0 owns what there is
    keys, wts = np.concatenate(keys), np.concatenate(wts)
    o = np.argsort(keys, kind="stable")
    keys, cum = keys[o], np.cumsum(wts[o])
    want = cum[-1] * np.arange(1, n_parts, dtype=np.float64) / n_parts
    at = np.minimum(np.searchsorted(cum, want, side="left"), keys.shape[0] - 1)
    return np.ascontiguousarray(keys[at], dtype=np.uint64)


def part_stem(directory, sample: int, part: int) -> str:
    import os
    return os.path.join(str(directory), f"sample{sample}.part{part}")


def write_parts(directory, sample: int, seqs, counts: np.ndarray, bounds: np.ndarray, iupac: bool = False):
    """the stretches of a range-split dictionary (``DeviceReads.range_split``) as one SampleReads file set per owner"""
    import os
    os.makedirs(str(directory), exist_ok=True)
    off = seqs.offsets
    ln = seqs.lengths
    mx = int(ln.max()) if len(ln) else 0
    ln = ln.astype(np.uint8 if mx < 256 else np.uint16 if mx < 65536 else np.int64)
    cnt = np.ascontiguousarray(counts, dtype=np.uint32).reshape(len(ln), -1)[:, 0]
    for q in range(len(bounds) - 1):
        a, b = int(bounds[q]), int(bounds[q + 1])
        stem = part_stem(directory, sample, q)
        np.save(stem + ".data.npy", seqs.data[int(off[a]):int(off[b])])
        np.save(stem + ".lengths.npy", ln[a:b])
        np.save(stem + ".counts.npy", cnt[a:b])


def read_part(directory, sample: int, part: int, remove: bool = True) -> "SampleReads":
    return SampleReads.from_files(part_stem(directory, sample, part) + "|0", remove)


def gather_tables(local: Sequence[SampleTables], rank: int, world: int, dist=None) -> Optional[List[SampleTables]]:
    """All ranks' SampleTables on rank 0 (None elsewhere), ordered by sample index."""
    if world == 1 or dist is None:
        return sorted(local, key=lambda t: t.index)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(list(local), gathered, dst=0)
    if rank != 0:
        return None
    flat = [t for part in gathered for t in part]
    return sorted(flat, key=lambda t: t.index)


def merge_tables(tables: Sequence[SampleTables]):
    """Per-sample columns side by side: (base_names, counters, class_sums[P,S], exact[R,S], iso[R,S])."""
    names = [t.name for t in tables]
    cls = np.stack([t.class_sums for t in tables], axis=1).astype(np.int64)
    ex = np.stack([t.exact for t in tables], axis=1).astype(np.int64)
    iso = np.stack([t.iso for t in tables], axis=1).astype(np.int64)
    src = {t.name: int(t.total_input) for t in tables}
    trimmed = {t.name: int(t.trimmed_all) for t in tables}
    uniq = {t.name: int(t.trimmed_unique) for t in tables}
    return names, src, trimmed, uniq, cls, ex, iso


def agree_on_run_directory(name: Optional[str], rank: int, world: int, dist=None) -> str:
    """The run's directory name, the same on every rank.  Without ``-onam`` it carries the start time to the second
    (mirge/__main__.py:69-75), and ranks that start either side of a second boundary would each make up their own: rank 0
    chooses, the others take its choice (one broadcast, right after the process group exists)."""
    import time
    if rank == 0 and not name:
        name = "miRge." + time.strftime('%Y-%m-%d_%H-%M-%S', time.localtime())
    if world == 1 or dist is None:
        return name
    box = [name if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def directory_is_shared(directory, rank: int, world: int, dist=None) -> bool:
    """True on a rank that sees the files rank 0 writes into ``directory`` (same node, or a shared filesystem): rank 0 drops
    a token file, everybody looks for it.  A rank that does not see it sends its dictionaries in-band (``gather_object``)
    instead of through files there."""
    import os
    import uuid
    if world == 1 or dist is None:
        return True
    box = [uuid.uuid4().hex if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    token = os.path.join(str(directory), ".mirge_probe_" + box[0])
    if rank == 0:
        os.makedirs(str(directory), exist_ok=True)
        with open(token, "w") as fh:
            fh.write("x")
    dist.barrier()
    seen = os.path.exists(token)
    dist.barrier()
    if rank == 0:
        os.unlink(token)
    return seen


def run_sharded(n_samples: int, rank: int, world: int, process: Callable[[int], SampleTables], dist=None):
    """Run ``process(i)`` for this rank's samples and gather on rank 0."""
    mine = assign_samples(n_samples, world)[rank]
    local = [process(i) for i in mine]
    return gather_tables(local, rank, world, dist)
