"""Flat (ragged) sequence containers shared by the host code, the tests and the bench.

A set of sequences is held as one ASCII byte array plus an int64 offsets array
(``offsets[i]:offsets[i+1]`` is sequence ``i``).  That is the exact shape the C ABI takes
(``include/mirge_native.h``), so nothing is re-laid-out between Python and the kernels.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Sequence

import numpy as np


@dataclass
class FlatSeqs:
    """Ragged ASCII sequences: ``data`` uint8, ``offsets`` int64 of length n+1."""
    data: np.ndarray
    offsets: np.ndarray

    def __len__(self) -> int:
        return int(self.offsets.shape[0] - 1)

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets)

    def get(self, i: int) -> str:
        return self.data[self.offsets[i]:self.offsets[i + 1]].tobytes().decode("ascii")

    def to_list(self) -> List[str]:
        buf = self.data.tobytes()
        o = self.offsets
        return [buf[o[i]:o[i + 1]].decode("ascii") for i in range(len(self))]

    @staticmethod
    def from_list(seqs: Sequence[str]) -> "FlatSeqs":
        lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
        offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        data = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8).copy()
        return FlatSeqs(data, offsets)

    @staticmethod
    def from_fixed(arr: np.ndarray) -> "FlatSeqs":
        """a numpy array of fixed-width byte strings ('S<k>', NUL-padded) -> ragged sequences without the padding"""
        arr = np.ascontiguousarray(arr)
        k = arr.dtype.itemsize
        mat = arr.view(np.uint8).reshape(-1, k)
        used = mat != 0
        lens = used.sum(axis=1).astype(np.int64)
        offsets = np.zeros(lens.shape[0] + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        return FlatSeqs(mat[used], offsets)

    @staticmethod
    def join_columns(cols: Sequence["FlatSeqs"], seps: bytes) -> bytes:
        """the text of a table: row i = cols[0][i] seps[0] cols[1][i] seps[1] ... (one separator byte after every
        column, the last one usually a newline), built with numpy scatters"""
        n = len(cols[0])
        assert len(seps) == len(cols) and all(len(c) == n for c in cols)
        lens = [c.lengths for c in cols]
        row_len = sum(lens) + len(cols)
        roff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(row_len, out=roff[1:])
        out = np.empty(int(roff[-1]), dtype=np.uint8)
        at = roff[:-1].copy()
        for c, ln, sep in zip(cols, lens, seps):
            rows = np.repeat(np.arange(n, dtype=np.int64), ln)
            out[at[rows] + (np.arange(c.data.shape[0], dtype=np.int64) - c.offsets[:-1][rows])] = c.data
            at = at + ln
            out[at] = sep
            at = at + 1
        return out.tobytes()

    def _ranges(self, starts: np.ndarray, lens: np.ndarray) -> "FlatSeqs":
        offsets = np.zeros(lens.shape[0] + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        rows = np.repeat(np.arange(lens.shape[0], dtype=np.int64), lens)
        within = np.arange(int(offsets[-1]), dtype=np.int64) - offsets[:-1][rows]
        return FlatSeqs(self.data[starts[rows] + within], offsets)

    def umi_split(self, front: int, back: int):
        """``UMIParser`` of every sequence (digest.py:305-315) -> (inserts ``s[f:-b]``, tags ``s[:f]+s[-b:]``);
        Python slice semantics, including ``s[-0:]`` = the whole sequence."""
        f, b = int(front), int(back)
        ln, o = self.lengths, self.offsets[:-1]
        lo = np.minimum(f, ln)
        hi = np.maximum(ln - b, 0) if b != 0 else ln
        pure = self._ranges(o + lo, np.maximum(hi - lo, 0))
        tail = np.minimum(b, ln) if b != 0 else ln
        head, end = self._ranges(o, lo), self._ranges(o + ln - tail, tail)
        tl = lo + tail
        toff = np.zeros(ln.shape[0] + 1, dtype=np.int64)
        np.cumsum(tl, out=toff[1:])
        tdata = np.empty(int(toff[-1]), np.uint8)
        rows = np.repeat(np.arange(ln.shape[0], dtype=np.int64), lo)
        tdata[toff[:-1][rows] + (np.arange(head.data.shape[0]) - head.offsets[:-1][rows])] = head.data
        rows = np.repeat(np.arange(ln.shape[0], dtype=np.int64), tail)
        tdata[toff[:-1][rows] + lo[rows] + (np.arange(end.data.shape[0]) - end.offsets[:-1][rows])] = end.data
        return pure, FlatSeqs(tdata, toff)

    def take(self, idx: np.ndarray) -> "FlatSeqs":
        idx = np.asarray(idx, dtype=np.int64)
        lens = self.lengths[idx]
        offsets = np.zeros(idx.shape[0] + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        total = int(offsets[-1])
        # gather: output byte j of row r comes from input byte j + (start of idx[r] in the input - start of r in the output):
        # ONE index array of the output's size (32-bit when the input allows: half the memory traffic), built in place
        it = np.int32 if max(total, int(self.offsets[-1])) < (1 << 31) - 1 else np.int64
        index = np.repeat((self.offsets[:-1][idx] - offsets[:-1]).astype(it), lens)
        index += np.arange(total, dtype=it)
        return FlatSeqs(self.data[index], offsets)


@dataclass
class Library:
    """One small-RNA reference library (what a bowtie index stands for in the reference,
    ``mirge/libs/manifoldAlign.py:84,97-98``).

    ``names`` are the SAM ``RNAME`` values bowtie would print, i.e. the FASTA header up to
    the first whitespace; ``headers`` keep the full header line (``bowtie-inspect -n``
    prints those, ``mirge/libs/summary.py:776-788``).
    """
    names: List[str]
    seqs: FlatSeqs
    headers: List[str] = field(default_factory=list)

    def __post_init__(self):
        if not self.headers:
            self.headers = list(self.names)

    def __len__(self) -> int:
        return len(self.names)

    @property
    def total_len(self) -> int:
        return int(self.seqs.offsets[-1])


def read_fasta(path: str) -> Library:
    names: List[str] = []
    headers: List[str] = []
    chunks: List[str] = []
    cur: List[str] = []
    with open(path, "r") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if line[0] == ">":
                if headers:
                    chunks.append("".join(cur))
                    cur = []
                headers.append(line[1:])
                names.append(line[1:].split()[0] if line[1:].split() else "")
            else:
                cur.append(line.strip().upper().replace("U", "T"))
    if headers:
        chunks.append("".join(cur))
    return Library(names, FlatSeqs.from_list(chunks), headers)


def write_fasta(path: str, lib: Library) -> None:
    seqs = lib.seqs.to_list()
    with open(path, "w") as fh:
        for h, s in zip(lib.headers, seqs):
            fh.write(">" + h + "\n" + s + "\n")


# Library keys in cascade order and the reference's index-name suffixes
# (mirge/libs/manifoldAlign.py:84,97,115-117).  ``{db}`` is appended for the miRNA and
# hairpin libraries only.
LIB_KEYS = ["mirna", "hairpin", "mature_trna", "pre_trna", "snorna", "rrna",
            "ncrna_others", "mrna", "spike-in"]
_DB_SUFFIXED = {"mirna", "hairpin"}


def index_basename(organism: str, key: str, ref_db: str) -> str:
    return f"{organism}_{key}_{ref_db}" if key in _DB_SUFFIXED else f"{organism}_{key}"


def load_library_dir(libraries_path: str, organism: str, ref_db: str,
                     with_spike: bool = False) -> Dict[str, Library]:
    """Load the libraries of ``<lib>/<org>/index.Libs/``: the bowtie-1 indexes a miRge3.0 library directory ships
    (``<org>_<key>[_<db>].{1,3,4}.ebwt``, read by ``ebwt.read_ebwt`` -- names from the tail of ``.1.ebwt``, sequences
    from ``.3.ebwt`` / ``.4.ebwt``), or ``<org>_<key>[_<db>].fa`` where a FASTA sits under the index's base name (it
    wins when both exist: it is what the index was built from and loads without decoding)."""
    base = os.path.join(libraries_path, organism, "index.Libs")
    out: Dict[str, Library] = {}
    for key in LIB_KEYS:
        if key == "spike-in" and not with_spike:
            continue
        out[key] = load_index(os.path.join(base, index_basename(organism, key, ref_db)))
    return out


def load_index(base: str, use_cache: bool = True) -> Library:
    """One library by its index base name: ``<base>.fa`` or ``<base>.{1,3,4}.ebwt[l]`` -- or the packed image of either
    from ``<base>.mirge3amd`` when that cache is current (libcache.py).  A library read from its source carries
    ``cache_target``: who packs it (``Cascade``) writes the cache."""
    from . import libcache
    if use_cache:
        lib = libcache.load(base)
        if lib is not None:
            return lib
    if os.path.exists(base + ".fa"):
        lib = read_fasta(base + ".fa")
    else:
        from . import ebwt
        if not ebwt.has_index(base):
            raise FileNotFoundError(f"library missing: neither {base}.fa nor {base}.1.ebwt exists")
        lib = ebwt.read_ebwt(base)
    if use_cache and libcache.enabled():
        lib.cache_target = (base, libcache.source_stamp(base))
    return lib


def load_merges(libraries_path: str, organism: str, ref_db: str) -> List[List[str]]:
    """``<org>_merges_<db>.csv``: first field merged name, rest members
    (mirge/libs/summary.py:707-712).  Missing file -> no merges (``:713-714``)."""
    p = os.path.join(libraries_path, organism, "annotation.Libs",
                     f"{organism}_merges_{ref_db}.csv")
    rows: List[List[str]] = []
    try:
        with open(p, "r") as fh:
            for line in fh:
                rows.append(line.strip().split(","))
    except FileNotFoundError:
        pass
    return rows
