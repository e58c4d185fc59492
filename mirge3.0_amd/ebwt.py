"""Reader for bowtie-1 index files (``<index>.1.ebwt`` / ``.3.ebwt`` / ``.4.ebwt``, and the 64-bit ``.ebwtl`` forms).

A miRge3.0 library directory holds only bowtie indexes; the reference recovers names and sequences by running
``bowtie-inspect`` (``mirge/libs/summary.py:776-788,812-827,1164-1175``, ``bamFmt.py:10-12``).  This module reads the
same information straight from the files, so that the library directory layout stays what it is without bowtie
installed:

* reference names: the tail of ``.1.ebwt`` -- one name per line, closed by a NUL byte (bowtie's
  ``Ebwt::writeFromMemory`` / ``readEbwtRefnames``).  The names are the full FASTA header lines; bowtie prints only the
  first whitespace-delimited token as SAM ``RNAME``.
* sequences: ``.3.ebwt`` = records ``(off, len, first)`` -- ``off`` ambiguous characters, then a stretch of ``len``
  unambiguous ones; ``first`` marks the first record of a reference -- and ``.4.ebwt`` = the unambiguous stretches
  2-bit packed, four bases per byte, base i at bits ``2*(i & 3)`` (bowtie's ``BitPairReference``).  ``bowtie-build``
  writes both files unless it is run with ``--noref``; an index without them is refused (decoding the BWT itself is not
  implemented).

Status: written from the bowtie 1.x sources' published format, round-tripped against ``tests/ebwt_writer.py`` (the
matching minimal writer, test infrastructure).  **Unverified against an index written by a real bowtie-build** -- none is
available offline; ``tools/bowtie_crosscheck.py --libs`` reads real indexes through this module when a box has them.
"""
from __future__ import annotations

import os
import struct
from typing import List, Tuple

import numpy as np

from .seqio import FlatSeqs, Library


def _suffix(base: str) -> Tuple[str, int]:
    """-> (extension, bytes per offset) of the index that exists for ``base``"""
    if os.path.exists(base + ".1.ebwt"):
        return ".ebwt", 4
    if os.path.exists(base + ".1.ebwtl"):
        return ".ebwtl", 8
    raise FileNotFoundError(f"no bowtie index {base}.1.ebwt[l]")


def has_index(base: str) -> bool:
    return os.path.exists(base + ".1.ebwt") or os.path.exists(base + ".1.ebwtl")


def read_header(base: str) -> dict:
    """The fixed-size head of ``.1.ebwt``: endianness word, text length, line / offset / ftab rates, flags, and the
    number of reference sequences that follows it."""
    ext, osz = _suffix(base)
    with open(base + ".1" + ext, "rb") as fh:
        head = fh.read(4 + osz + 5 * 4 + osz)
    one_le = struct.unpack("<i", head[:4])[0]
    if one_le == 1:
        en = "<"
    elif struct.unpack(">i", head[:4])[0] == 1:
        en = ">"
    else:
        raise ValueError(f"{base}.1{ext}: not a bowtie index (endianness word {head[:4]!r})")
    o = "I" if osz == 4 else "Q"
    length, = struct.unpack(en + o, head[4:4 + osz])
    line_rate, lines_per_side, off_rate, ftab_chars, flags = struct.unpack(en + "5i", head[4 + osz:4 + osz + 20])
    n_pat, = struct.unpack(en + o, head[4 + osz + 20:])
    return dict(endian=en, off_size=osz, ext=ext, len=length, lineRate=line_rate, linesPerSide=lines_per_side,
                offRate=off_rate, ftabChars=ftab_chars, flags=flags, nPat=n_pat,
                color=bool(flags < 0 and (-flags) & 2), entireReverse=bool(flags < 0 and (-flags) & 4))


def names_offset(h: dict, n_frag: int) -> int:
    """Byte offset of the name block in ``.1.ebwt`` as bowtie's ``EbwtParams`` lays the file out (used to cross-check
    the tail scan of ``read_names``; ``n_frag`` is read from the file)."""
    osz = h["off_size"]
    bwt_sz = h["len"] // 4 + 1
    side_sz = (1 << h["lineRate"]) * max(h["linesPerSide"], 1)
    side_bwt_sz = side_sz - 2 * osz
    n_side_pairs = (bwt_sz + 2 * side_bwt_sz - 1) // (2 * side_bwt_sz)
    ebwt_tot = n_side_pairs * 2 * side_sz
    ftab_len = (1 << (2 * h["ftabChars"])) + 1
    eftab_len = 2 * h["ftabChars"]
    return (4 + osz + 20) + osz * (1 + h["nPat"]) + osz * (1 + 3 * n_frag) + ebwt_tot + osz + osz * (5 + ftab_len + eftab_len)


def _printable(b: bytes) -> bool:
    return all(0x20 <= c <= 0x7e or c == 0x09 for c in b)


def read_names(base: str) -> List[str]:
    """Reference names (full header lines) from the tail of ``.1.ebwt``: nPat newline-terminated strings closed by a
    NUL.  The block's offset follows from the header (``names_offset``); if the file does not have printable names
    there (a layout this reader does not know), the last nPat lines of the file are taken instead and binary bytes
    in front of the first one are dropped."""
    h = read_header(base)
    n = int(h["nPat"])
    if n == 0:
        return []
    osz, en = h["off_size"], h["endian"]
    path = base + ".1" + h["ext"]
    size = os.path.getsize(path)
    with open(path, "rb") as fh:
        fh.seek(4 + osz + 20 + osz * (1 + n))
        w = fh.read(osz)
        n_frag = struct.unpack(en + ("I" if osz == 4 else "Q"), w)[0] if len(w) == osz else 0
        at = names_offset(h, n_frag)
        if 0 < at < size:
            fh.seek(at)
            parts = _name_lines(fh.read())
            if len(parts) == n and all(_printable(p) for p in parts):
                return [p.decode("ascii") for p in parts]
        chunk = min(size, 1 << 16)
        while True:
            fh.seek(size - chunk)
            parts = _name_lines(fh.read(chunk))
            if len(parts) > n or chunk == size:
                break
            chunk = min(size, chunk * 4)
    if len(parts) < n:
        raise ValueError(f"{path}: {n} reference names expected, {len(parts)} found at the end of the file")
    parts = parts[len(parts) - n:]
    head = parts[0]
    k = len(head)
    while k > 0 and (0x20 <= head[k - 1] <= 0x7e or head[k - 1] == 0x09):
        k -= 1
    parts[0] = head[k:]
    return [p.decode("ascii", "replace") for p in parts]


def _name_lines(tail: bytes) -> List[bytes]:
    body = tail[:-1] if tail.endswith(b"\0") else tail
    if body.endswith(b"\n"):
        body = body[:-1]
    return body.split(b"\n")


def read_sequences(base: str) -> Tuple[FlatSeqs, List[int]]:
    """-> (sequences, records per sequence) decoded from ``.3.ebwt`` (records) + ``.4.ebwt`` (2-bit bases)."""
    ext, osz = _suffix(base)
    p3, p4 = base + ".3" + ext, base + ".4" + ext
    if not (os.path.exists(p3) and os.path.exists(p4)):
        raise FileNotFoundError(f"{base}: the index has no {os.path.basename(p3)} / {os.path.basename(p4)} (built with --noref?); "
                                "the sequences cannot be recovered without them")
    raw = np.fromfile(p3, dtype=np.uint8)
    if raw.size < 8:
        raise ValueError(f"{p3}: truncated")
    en = "<" if struct.unpack("<i", raw[:4].tobytes())[0] == 1 else ">"
    if struct.unpack(en + "i", raw[:4].tobytes())[0] != 1:
        raise ValueError(f"{p3}: not a bowtie reference file")
    # the record count is one index offset wide (TIndexOffU: 32 bits in .ebwt, 64 in .ebwtl), like the records' fields
    nrec = int(np.frombuffer(raw[4:4 + osz].tobytes(), dtype=np.dtype(en + ("u4" if osz == 4 else "u8")))[0])
    rec_sz = 2 * osz + 1
    body = raw[4 + osz:]
    if body.size < nrec * rec_sz:
        raise ValueError(f"{p3}: {nrec} records announced, {body.size // rec_sz} present")
    body = body[:nrec * rec_sz].reshape(nrec, rec_sz)
    dt = np.dtype(en + ("u4" if osz == 4 else "u8"))
    off = np.ascontiguousarray(body[:, :osz]).view(dt).reshape(nrec).astype(np.int64)
    ln = np.ascontiguousarray(body[:, osz:2 * osz]).view(dt).reshape(nrec).astype(np.int64)
    first = body[:, 2 * osz] != 0
    if nrec and not first[0]:
        raise ValueError(f"{p3}: the first record is not marked as the start of a reference")
    packed = np.fromfile(p4, dtype=np.uint8)
    total = int(ln.sum())
    if packed.size * 4 < total:
        raise ValueError(f"{p4}: {total} bases announced by {p3}, {packed.size * 4} present")
    # unpack the 2-bit stream four bases at a time (one table look-up per packed byte), then lay the stretches out with their
    # runs of N in front: the output is [off_0 x N][ln_0 bases][off_1 x N][ln_1 bases]... in record order, so one boolean mask
    # of the output's length places every base -- no per-base index arrays (a 130 Mb index took 22 s with them, 1 s without)
    lut = np.empty(256, dtype="<u4")
    b4 = np.frombuffer(b"ACGT", dtype=np.uint8).astype(np.uint32)
    v = np.arange(256, dtype=np.uint32)
    lut[:] = b4[v & 3] | (b4[(v >> 2) & 3] << 8) | (b4[(v >> 4) & 3] << 16) | (b4[(v >> 6) & 3] << 24)
    bases = lut[packed[:(total + 3) // 4]].view(np.uint8)[:total]
    rec_total = off + ln
    seq_id = np.cumsum(first) - 1
    n_seq = int(seq_id[-1] + 1) if nrec else 0
    seq_len = np.bincount(seq_id, weights=rec_total, minlength=n_seq).astype(np.int64) if nrec else np.zeros(0, np.int64)
    offsets = np.zeros(n_seq + 1, dtype=np.int64)
    np.cumsum(seq_len, out=offsets[1:])
    if nrec and not off.any():
        data = np.ascontiguousarray(bases)  # no ambiguous stretch anywhere: the stream is the text
    else:
        data = np.full(int(offsets[-1]), ord("N"), dtype=np.uint8)
        if nrec:
            runs = np.empty(2 * nrec, dtype=np.int64)
            runs[0::2], runs[1::2] = off, ln
            is_base = np.repeat(np.tile(np.array([False, True]), nrec), runs)
            data[is_base] = bases
    recs_per_seq = np.bincount(seq_id, minlength=n_seq).tolist() if nrec else []
    return FlatSeqs(data, offsets), recs_per_seq


def read_ebwt(base: str) -> Library:
    """``<base>.{1,3,4}.ebwt`` -> Library (names = first token of each header, headers = the stored names)."""
    headers = read_names(base)
    seqs, _ = read_sequences(base)
    if len(headers) != len(seqs):
        raise ValueError(f"{base}: {len(headers)} names in .1.ebwt but {len(seqs)} sequences in .3.ebwt")
    names = [(h.split()[0] if h.split() else "") for h in headers]
    return Library(names, seqs, headers)
