"""Library cache: the one-time conversion of a library index kept next to it (hard part 2).

The reference opens a bowtie index in every pass (``manifoldAlign.py:97-99``); the MI355X engine wants the references as a
2-bit text + invalid-base bitmap + reference starts in HBM.  Getting there from ``<index>.fa`` (parse) or
``<index>.{1,3,4}.ebwt`` (decode) and packing it is what a one-sample run used to wait for before its first kernel
(0.3 s of a 0.7 s run on the human-sized set).  ``<index>.mirge3amd`` holds that image together with the names: one
flat file, arrays memory-mapped on load.  It is written the first time a library is packed and ignored -- then rewritten --
whenever the index files' sizes or modification times differ from the ones it was made from, or its layout version does.
A library directory that cannot be written to gets its cache under ``$XDG_CACHE_HOME/mirge3_amd`` (or ``~/.cache``);
``MIRGE_LIB_CACHE=0`` switches the cache off.
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Dict, List, Optional

import numpy as np

from .seqio import FlatSeqs, Library

MAGIC = b"MIRGE3AMD-LIB\0\0\0"
VERSION = 3  # 3: a 64-bit content hash per array in the header, verified on load
ASCII_LIMIT = 8 << 20  # libraries up to this many bases also keep their letters (IUPAC codes, for the host-side reports)


def content_hash(a: np.ndarray, algo: Optional[str] = None):
    """(algorithm, 64-bit hash as hex) of an array's bytes.  xxh3_64 where the xxhash module exists (~10 GB/s: the 55 MB image
    of the human-sized set in ~5 ms), else blake2b with an 8-byte digest.  `algo` forces the one a cache was written with;
    None when that one is not available here (the cache then counts as unverifiable and is rebuilt)."""
    buf = memoryview(np.ascontiguousarray(a)).cast("B")
    if algo in (None, "xxh3_64"):
        try:
            import xxhash
            return "xxh3_64", xxhash.xxh3_64(buf).hexdigest()
        except ImportError:
            if algo is not None:
                return None
    if algo in (None, "blake2b8"):
        return "blake2b8", hashlib.blake2b(buf, digest_size=8).hexdigest()
    return None


def enabled() -> bool:
    return os.environ.get("MIRGE_LIB_CACHE", "1") != "0"


def source_files(base: str) -> List[str]:
    if os.path.exists(base + ".fa"):
        return [base + ".fa"]
    out = []
    for ext in (".1.ebwt", ".3.ebwt", ".4.ebwt", ".1.ebwtl", ".3.ebwtl", ".4.ebwtl"):
        if os.path.exists(base + ext):
            out.append(base + ext)
    return out


def source_stamp(base: str):
    return [[os.path.basename(f), os.path.getsize(f), os.stat(f).st_mtime_ns] for f in source_files(base)]


def cache_paths(base: str) -> List[str]:
    """where the cache of an index may live: next to it, else in the user's cache directory"""
    home = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    tag = hashlib.sha1(os.path.abspath(base).encode()).hexdigest()[:16]
    return [base + ".mirge3amd", os.path.join(home, "mirge3_amd", tag + "_" + os.path.basename(base) + ".mirge3amd")]


class PackedSeqs(FlatSeqs):
    """The sequences of a cached library: lengths at once, letters decoded from the 2-bit image only if somebody asks
    (the engine takes the image itself, ``mirge_lib_create_packed``)."""

    def __init__(self, offsets: np.ndarray, packed: Dict[str, np.ndarray], ascii_data: Optional[np.ndarray] = None):
        self.offsets = offsets
        self.packed = packed
        self._data = ascii_data

    @property
    def data(self) -> np.ndarray:  # type: ignore[override]
        if self._data is None:
            self._data = decode(self.packed, self.offsets)
        return self._data


def decode(packed: Dict[str, np.ndarray], offsets: np.ndarray) -> np.ndarray:
    """2-bit text + invalid bitmap -> ASCII (A/C/G/T, N where the bitmap says so), separators dropped"""
    total = int(packed["total"])
    lens = np.diff(offsets)
    out = np.empty(int(offsets[-1]), dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    T, inv, rs = packed["T"], packed["inv"], packed["ref_start"]
    step = 1 << 22
    # reference t occupies global positions [rs[t], rs[t] + len): walk the text in chunks, copy what lies inside references
    g0 = 0
    t = 0
    n = lens.shape[0]
    while g0 < total and t < n:
        g1 = min(total, g0 + step)
        g = np.arange(g0, g1, dtype=np.int64)
        code = ((T[g >> 5] >> ((g & 31) * 2).astype(np.uint64)) & np.uint64(3)).astype(np.int64)
        bad = ((inv[g >> 6] >> (g & 63).astype(np.uint64)) & np.uint64(1)).astype(bool)
        ch = np.where(bad, np.uint8(ord("N")), letters[code])
        ref = np.searchsorted(rs, g, side="right") - 1
        within = g - rs[ref].astype(np.int64)
        ok = (ref < n) & (within < lens[np.minimum(ref, n - 1)])
        out[(offsets[:-1][np.minimum(ref, n - 1)] + within)[ok]] = ch[ok]
        g0 = g1
    return out


def save(base: str, lib: Library, packed: Dict[str, np.ndarray], stamp) -> Optional[str]:
    """write ``<base>.mirge3amd`` (or the user-cache twin); None when neither place can be written"""
    from .seqio import FlatSeqs as _FS
    names = _FS.from_list(lib.names)
    headers = _FS.from_list(lib.headers)
    arrays = {
        "T": np.ascontiguousarray(packed["T"], dtype=np.uint64), "inv": np.ascontiguousarray(packed["inv"], dtype=np.uint64),
        "ref_start": np.ascontiguousarray(packed["ref_start"], dtype=np.uint32),
        "seq_offsets": np.ascontiguousarray(lib.seqs.offsets, dtype=np.int64),
        "names_data": names.data, "names_off": names.offsets, "headers_data": headers.data, "headers_off": headers.offsets,
    }
    if int(lib.seqs.offsets[-1]) <= ASCII_LIMIT:
        arrays["ascii"] = np.ascontiguousarray(lib.seqs.data, dtype=np.uint8)
    meta = {"version": VERSION, "source": stamp, "total": int(packed["total"]), "kmax": int(packed["kmax"]),
            "valid_positions": int(packed["valid_positions"]), "n_refs": len(lib), "arrays": {}}
    at = 0
    for k, a in arrays.items():
        at = (at + 63) & ~63
        algo, digest = content_hash(a)
        meta["arrays"][k] = {"dtype": a.dtype.str, "n": int(a.shape[0]), "at": at, "hash": digest, "hash_algo": algo}
        at += a.nbytes
    head = json.dumps(meta).encode()
    body0 = (len(MAGIC) + 8 + len(head) + 63) & ~63
    for path in cache_paths(base):
        tmp = None
        try:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)  # a bare index name: the current directory
            tmp = path + f".tmp{os.getpid()}"
            with open(tmp, "wb") as fh:
                fh.write(MAGIC + len(head).to_bytes(8, "little") + head)
                fh.write(b"\0" * (body0 - fh.tell()))
                for k, a in arrays.items():
                    fh.write(b"\0" * (body0 + meta["arrays"][k]["at"] - fh.tell()))
                    fh.write(a.tobytes() if a.nbytes < (1 << 20) else memoryview(a).cast("B"))
            os.replace(tmp, path)  # readers see the old cache or the new one, never a half-written file
            return path
        except OSError:
            if tmp is not None:
                try:
                    os.unlink(tmp)
                except OSError:
                    pass
    return None


def load(base: str) -> Optional[Library]:
    """the cached library of an index, or None (no cache, other layout version, index files changed since, or an array whose
    bytes no longer hash to what the header says: a damaged file next to a shared library directory must not turn into wrong
    annotation -- it is ignored, and the caller's next save() replaces it)"""
    if not enabled():
        return None
    try:
        stamp = source_stamp(base)
    except OSError:
        return None
    if not stamp:
        return None
    for path in cache_paths(base):
        try:
            with open(path, "rb") as fh:
                if fh.read(len(MAGIC)) != MAGIC:
                    continue
                hl = int.from_bytes(fh.read(8), "little")
                meta = json.loads(fh.read(hl))
            if meta.get("version") != VERSION or meta.get("source") != stamp:
                continue
            body0 = (len(MAGIC) + 8 + hl + 63) & ~63
            arr = {}
            for k, d in meta["arrays"].items():
                arr[k] = np.memmap(path, dtype=np.dtype(d["dtype"]), mode="r", offset=body0 + d["at"], shape=(d["n"],)) if d["n"] else \
                    np.zeros(0, dtype=np.dtype(d["dtype"]))
                if content_hash(arr[k], d["hash_algo"]) != (d["hash_algo"], d["hash"]):
                    raise ValueError(f"{path}: array {k} does not match its content hash")
            packed = {"T": arr["T"], "inv": arr["inv"], "ref_start": arr["ref_start"], "total": meta["total"],
                      "kmax": meta["kmax"], "valid_positions": meta["valid_positions"]}
            names = FlatSeqs(np.asarray(arr["names_data"]), np.asarray(arr["names_off"])).to_list()
            headers = FlatSeqs(np.asarray(arr["headers_data"]), np.asarray(arr["headers_off"])).to_list()
            seqs = PackedSeqs(np.asarray(arr["seq_offsets"]), packed, np.asarray(arr["ascii"]) if "ascii" in arr else None)
            lib = Library(names, seqs, headers)
            lib.cache_path = path
            return lib
        except (OSError, ValueError, KeyError):
            continue
    return None
