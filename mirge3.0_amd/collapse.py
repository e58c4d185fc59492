"""Read collapse and the sample matrix -- host mirror of ``mirge/libs/digest.py:105-302``.

``baking(args, inFileArray, inFileBaseArray, workDir)`` keeps the reference's signature and
return value ``(DataFrame, sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique)``.
The per-chunk dict counting (``cutadapt()``, digest.py:320-375), the dict merge (:141-163) and
the pandas outer join of the samples (:237-245) are ONE ``mirge_collapse`` call on the GPU: all
samples' reads go in together with a sample id and come back as the U distinct sequences plus a
U x S count matrix.

The cutadapt modifier chain of ``stipulate`` (digest.py:59-101) runs on the GPU too (``trim_from_args`` ->
``mirge_reads_parse_trim``: quality / NextSeq trimming, one 3' or one 5' adapter, N ends, cuts); what is not covered
(more than two adapters, ``--action mask`` / ``lowercase``) raises instead of being skipped.

UMI handling (digest.py:164-205,305-315,334-365) is part of the same device-resident parse
(``mirge_reads_parse_umi``): ``-umi f,b`` slices f bases off the front and b off the back of every counted read
(counts add up); with ``-udd`` the UMI-tagged reads are collapsed first, ``<sample>_umiCounts.csv`` is written from that
result and ONE insert per distinct tagged read goes on, so a count is a number of molecules; ``--qiagenumi`` takes the
UMI from behind the 3' adapter of the untrimmed read (the reference's ``currentSeq.split(trimmed)[1]`` string rule).
Both command lines of the reference's quick start (``docs/source/quick_start.md:282-314``) run this way.
``-tcf`` writes ``<sample>.trim.collapse.fa`` (digest.py:219-229).
"""
from __future__ import annotations

import gzip
import time
from pathlib import Path
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import PASS_COLUMNS, _ffi
from .seqio import FlatSeqs


def read_fastq_sequences(path: str) -> FlatSeqs:
    """Sequence lines of a FASTQ (or FASTA, or one-sequence-per-line) file, plain or .gz."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as fh:
        buf = np.frombuffer(fh.read(), dtype=np.uint8)
    if buf.size == 0:
        return FlatSeqs(np.zeros(0, np.uint8), np.zeros(1, np.int64))
    nl = np.flatnonzero(buf == 10)
    if buf[-1] != 10:
        nl = np.append(nl, buf.size)
    starts = np.concatenate(([0], nl[:-1] + 1))
    ends = nl.copy()
    # strip \r
    cr = (ends > starts) & (buf[np.maximum(ends - 1, 0)] == 13)
    ends = ends - cr
    first = buf[0]
    if first == ord("@"):
        sel = slice(1, None, 4)
    elif first == ord(">"):
        sel = slice(1, None, 2)
    else:
        sel = slice(0, None, 1)
    s, e = starts[sel], ends[sel]
    lens = (e - s).astype(np.int64)
    offsets = np.zeros(lens.shape[0] + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    rows = np.repeat(np.arange(lens.shape[0], dtype=np.int64), lens)
    within = np.arange(int(offsets[-1]), dtype=np.int64) - offsets[:-1][rows]
    return FlatSeqs(buf[s[rows] + within], offsets)


def read_text(path: str) -> bytes:
    """The file's bytes (gunzipped if .gz) for the device-side parser (``mirge_reads_parse``).  A FASTA whose sequences
    are wrapped over several lines (dnaio reads those) is unwrapped here: the device parser finds records by line
    number and refuses anything else."""
    if str(path).endswith(".gz"):
        with gzip.open(path, "rb") as fh:
            data = fh.read()
    else:
        import os
        if os.path.getsize(path) == 0:
            return b""
        data = np.memmap(path, dtype=np.uint8, mode="r")  # the pages go from the page cache to the GPU: no read() copy
    if bytes(data[:1]) == b">":
        data = unwrap_fasta(bytes(data))
    return data


def read_texts(paths, depth: int = 4):
    """``read_text`` of every file, in order, read ahead by up to ``depth`` files on worker threads: gunzipping a sample
    (zlib releases the GIL) then runs beside the previous sample's transfer and parse, and several .gz samples inflate
    on several cores -- a single inflate stream is the slowest stage of a run from compressed FASTQ."""
    from concurrent.futures import ThreadPoolExecutor
    paths = [str(p) for p in paths]
    if len(paths) <= 1:
        for p in paths:
            yield read_text(p)
        return
    with ThreadPoolExecutor(max_workers=max(1, min(depth, len(paths)))) as pool:
        pending = [pool.submit(read_text, p) for p in paths[:depth]]
        nxt = len(pending)
        while pending:
            text = pending.pop(0).result()
            if nxt < len(paths):
                pending.append(pool.submit(read_text, paths[nxt]))
                nxt += 1
            yield text


def unwrap_fasta(data: bytes) -> bytes:
    """FASTA text -> one sequence line per record.  A newline stays when it ends a header line or when the next line
    is a header (or the text ends); every other newline (and '\\r') sits inside a sequence and is dropped."""
    buf = np.frombuffer(data, dtype=np.uint8)
    nl = np.flatnonzero(buf == 10)
    if nl.size == 0:
        return data
    line_start = np.concatenate(([0], nl[:-1] + 1))
    is_header = buf[np.minimum(line_start, buf.size - 1)] == ord(">")
    nxt = nl + 1
    next_is_header = (nxt >= buf.size) | (buf[np.minimum(nxt, buf.size - 1)] == ord(">"))
    drop_nl = nl[~(is_header | next_is_header)]
    if drop_nl.size == 0:
        return data
    keep = np.ones(buf.size, dtype=bool)
    keep[drop_nl] = False
    cr = drop_nl[(drop_nl > 0)] - 1
    keep[cr[buf[cr] == 13]] = False
    return buf[keep].tobytes()


ILLUMINA_3P = 'TGGAATTCTCGGGTGCCAAGGAACTCCAG'  # what `-a illumina` stands for (mirge/__main__.py:65-83)
ILLUMINA_5P = 'GTTCAGAGTTCTACAGTCCGACGATC'     # what `-g illumina` stands for


def adapters_from_args(args):
    """``args.adapters`` as the reference has it after mirge/__main__.py:65-83: a list of (kind, sequence) in command-line
    order, 'illumina' spelled out -- with ONE adapter by its kind, with TWO by its position (the first is taken for the 3'
    adapter, the second for the 5' one, whatever their flags were: the reference's own rule).  Plain strings (3' adapters)
    and a separate ``front`` list are accepted from callers that build the namespace themselves."""
    adapters = getattr(args, "adapters", None) or []
    if isinstance(adapters, str):
        adapters = [adapters]
    adapters = [tuple(a) if isinstance(a, (tuple, list)) else ("back", a) for a in adapters]
    front = getattr(args, "front", None) or []
    if isinstance(front, str):
        front = [front]
    adapters += [tuple(f) if isinstance(f, (tuple, list)) else ("front", f) for f in front]
    if len(adapters) == 2:
        adapters = [(k, (ILLUMINA_3P, ILLUMINA_5P)[i] if q == "illumina" else q) for i, (k, q) in enumerate(adapters)]
    elif len(adapters) == 1:
        adapters = [(k, {"back": ILLUMINA_3P, "front": ILLUMINA_5P}.get(k, q) if q == "illumina" else q) for k, q in adapters]
    return adapters


def trim_from_args(args):
    """The cutadapt modifier chain of ``stipulate`` (digest.py:59-101) as the options of ``mirge_reads_parse_trim``:
    ``-q`` (default "10": quality trimming is ALWAYS in the reference's chain), ``-a`` / ``-g`` (one or two adapters, 3'
    or 5'; with two, a read loses the better match -- AdapterCutter with times = 1), ``-nxt``, ``-NX``, ``-u``,
    ``--overlap``, ``--error-rate``, ``-phr``, ``-n`` (repeat the removal), ``--no-indels``, ``--match-read-wildcards``, ``-N``,
    ``--action none``.  More than two adapters and ``--action mask`` / ``lowercase`` are refused."""
    adapters = adapters_from_args(args)
    if len(adapters) > 2 or any(kind not in ("back", "front") for kind, _ in adapters):
        raise NotImplementedError("up to two adapters are supported (-a / -g, in any combination)")
    q = getattr(args, "quality_cutoff", "10")
    qf, qb = 0, -1
    if q is not None:
        vals = [int(v) for v in str(q).split(",")]
        if len(vals) == 1:
            qf, qb = 0, vals[0]
        elif len(vals) == 2:
            qf, qb = vals
        else:
            raise SystemExit("Expected one value or two values separated by comma for the quality cutoff")
    cut = [int(c) for c in (getattr(args, "cut", None) or [])]
    if len(cut) > 2:
        raise SystemExit("You cannot remove bases from more than two ends.")
    if len(cut) == 2 and cut[0] * cut[1] > 0:
        raise SystemExit("You cannot remove bases from the same end twice.")
    nxt = getattr(args, "nextseq_trim", None)
    base = 64 if int(getattr(args, "phred64", 33) or 33) == 64 else 33
    a1 = adapters[0] if adapters else (None, None)
    a2 = adapters[1] if len(adapters) > 1 else (None, None)
    return _ffi.MirgeTrim.make(adapter=a1[1], quality_back=qb, quality_front=qf, nextseq=-1 if nxt is None else int(nxt),
                               phred_base=base, min_overlap=int(getattr(args, "overlap", 3)),
                               error_rate=float(getattr(args, "error_rate", 0.12)), trim_n=bool(getattr(args, "trim_n", False)),
                               cut=cut, count_per_modifier=getattr(args, "trim_count", "per-modifier") != "once",
                               front=a1[0] == "front", adapter2=a2[1], front2=a2[0] == "front",
                               times=int(getattr(args, "times", 1) or 1), indels=bool(getattr(args, "indels", True)),
                               read_wildcards=bool(getattr(args, "match_read_wildcards", False)),
                               adapter_wildcards=bool(getattr(args, "match_adapter_wildcards", True)),
                               action=str(getattr(args, "action", "trim") or "trim"))


def filter_min_length(reads: FlatSeqs, min_len: int) -> FlatSeqs:
    keep = np.flatnonzero(reads.lengths >= int(min_len))
    if keep.shape[0] == len(reads):
        return reads
    return reads.take(keep)


def collapse_samples(ctx: _ffi.Context, samples: Sequence[FlatSeqs]):
    """Host sequences of several samples -> the DeviceReads of their joint collapse (U unique reads + a U x S count
    matrix); the route for callers that hold sequences, not files (``baking`` parses the files' text on the GPU)."""
    S = len(samples)
    if S == 1:
        allr, sid = samples[0], None
    else:
        data = np.concatenate([s.data for s in samples])
        lens = np.concatenate([s.lengths for s in samples])
        off = np.zeros(lens.shape[0] + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        allr = FlatSeqs(data, off)
        sid = np.repeat(np.arange(S, dtype=np.int32), [len(s) for s in samples])
    raw = _ffi.DeviceReads.pack(ctx, allr)
    uniq = raw.collapse(sid, S)
    raw.close()
    return uniq


def umi_from_args(args):
    """``-umi f,b`` / ``--qiagenumi`` / ``-udd`` as the options of ``mirge_reads_parse_umi``, or None.  The reference reads
    ``umi.split(",")`` everywhere (digest.py:149,167,344,358): two integers."""
    umi = getattr(args, "uniq_mol_ids", None)
    qia = bool(getattr(args, "qiagenumi", False))
    if not umi:
        if qia:
            raise SystemExit("--qiagenumi requires -umi x,y (mirge/libs/parse.py: '-umi x,y Required')")
        return None
    try:
        f, b = (int(x) for x in str(umi).split(","))
    except ValueError:
        raise SystemExit("-umi expects two comma separated integers, e.g. 4,4 or 0,12")
    return _ffi.MirgeUmi.make(f, b, qiagen=qia, dedup=bool(getattr(args, "umiDedup", False)))


def write_umi_counts(path, tagged: "_ffi.DeviceReads", front: int, back: int, min_len: int) -> int:
    """``<sample>_umiCounts.csv`` (digest.py:183-197): one line ``UMI,insert,count`` per distinct UMI-tagged read whose insert
    has ``min_len`` bases, in the order the tagged reads first appeared; appended to, as the reference opens it.  Returns
    the number of lines.  Formatted with numpy from the flat arrays: no per-read Python object."""
    counts, _ = tagged.counts()
    full = tagged.unpack().take(tagged.first_appearance_order())
    c = counts[tagged.first_appearance_order(), 0] if len(tagged) else np.zeros(0, np.uint32)
    pure, tag = full.umi_split(front, back)
    keep = np.flatnonzero(pure.lengths >= int(min_len))
    pure, tag, c = pure.take(keep), tag.take(keep), c[keep]
    digits = FlatSeqs.from_fixed(np.char.mod("%d", c.astype(np.int64)).astype("S")) if len(c) else FlatSeqs.from_list([])
    with open(path, "ab") as fh:
        fh.write(b"UMISeq,transcriptSeq,UMICounts\n")
        fh.write(FlatSeqs.join_columns([tag, pure, digits], b",,\n"))
    return int(keep.shape[0])


def write_tcf(path, raw: "_ffi.DeviceReads") -> None:
    """``<sample>.trim.collapse.fa`` (``-tcf``, digest.py:219-229): the sample's own dictionary, most frequent first (Python's
    stable ``sorted(..., key=count, reverse=True)``: ties stay in dictionary order), ``>seq<k>_<count>``.  One collapse of the
    sample's raw reads on the GPU, the text assembled with numpy."""
    u1 = raw.collapse()
    cnt, first = u1.counts()
    c = cnt[:, 0].astype(np.int64) if len(u1) else np.zeros(0, np.int64)
    order = np.lexsort((first, -c))
    seqs = u1.unpack().take(order)
    u1.close()
    n = len(order)
    head = FlatSeqs.from_fixed(np.char.add(b">seq", np.char.mod("%d", np.arange(1, n + 1)).astype("S")).astype("S")) if n else FlatSeqs.from_list([])
    num = FlatSeqs.from_fixed(np.char.mod("%d", c[order]).astype("S")) if n else FlatSeqs.from_list([])
    with open(path, "wb") as fo:
        fo.write(FlatSeqs.join_columns([head, num, seqs], b"_\n\n"))


def parse_sample(ctx: _ffi.Context, text, min_len: int, trim, umi, workDir=None, name=None):
    """One file's text -> (raw reads as the collapse takes them, records seen): parse, the modifier chain, the length
    filter and -- with ``umi`` -- the reference's UMI handling, all on the GPU (``mirge_reads_parse[_trim|_umi]``).
    With ``-udd`` also writes ``<name>_umiCounts.csv``."""
    if umi is None:
        return _ffi.DeviceReads.parse(ctx, text, 0, min_len, trim)
    raw, n_rec, tagged = _ffi.DeviceReads.parse_umi(ctx, text, 0, min_len, trim, umi)
    if tagged is not None:
        if workDir is not None:
            write_umi_counts(Path(workDir) / (str(name) + "_umiCounts.csv"), tagged, umi.front, umi.back, min_len)
        tagged.close()
    return raw, n_rec


def baking(args, inFileArray, inFileBaseArray, workDir, ctx: _ffi.Context = None):
    """Drop-in for ``baking`` (digest.py:105-302): every file's text goes to the GPU as it is -- records found, trimmed,
    UMI-sliced, filtered by length and packed there (digest.py:320-375 without a per-read host step); the samples' read
    sets are appended on the device and collapsed together with a sample id (digest.py:141-163 + the outer join :243)."""
    import pandas as pd
    trim = trim_from_args(args)
    umi = umi_from_args(args)
    begningTime = time.perf_counter()
    runlogFile = Path(workDir) / "run.log"
    outlog = open(str(runlogFile), "a+")
    ctx = ctx or _ffi.Context(getattr(args, "device", 0))
    min_len = int(getattr(args, "minimum_length", 16))
    sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique = {}, {}, {}
    parsed: List[_ffi.DeviceReads] = []
    texts = read_texts(inFileArray)  # read (and gunzipped) ahead on worker threads
    for FQfile, name in zip(inFileArray, inFileBaseArray):
        start = time.perf_counter()
        raw, n_rec = parse_sample(ctx, next(texts), min_len, trim, umi, workDir, name)
        sampleReadCounts[name] = n_rec
        trimmedReadCounts[name] = len(raw)
        parsed.append(raw)
        finish2 = time.perf_counter()
        if not args.quiet:
            print(f'Cutadapt finished for file {name} in {round(finish2-start, 4)} second(s)')
        outlog.write(f'Cutadapt finished for file {name} in {round(finish2-start, 4)} second(s)\n')
        if getattr(args, "tcf_out", False):  # (:219-229): by count, ties in dict order
            write_tcf(Path(workDir) / (str(name) + '.trim.collapse.fa'), raw)
    t0 = time.perf_counter()
    if len(parsed) == 1:
        uniq = parsed[0].collapse()
    else:
        allr = _ffi.DeviceReads.concat(ctx, parsed)
        sid = np.repeat(np.arange(len(parsed), dtype=np.int32), [len(p) for p in parsed])
        uniq = allr.collapse(sid, len(parsed))
        allr.close()
    for p in parsed:
        p.close()
    counts, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    uniq.close()
    counts = counts.astype(np.int64)
    for s, name in enumerate(inFileBaseArray):
        trimmedReadCountsUnique[name] = int((counts[:, s] > 0).sum())
    if len(inFileBaseArray) == 1:
        order = np.argsort(first, kind="stable")  # dict insertion order
    else:
        order = np.argsort(np.asarray(seqs, dtype=object), kind="stable")  # outer join sorts the index
    complete_set = pd.DataFrame(counts[order], columns=list(inFileBaseArray),
                                index=pd.Index([seqs[i] for i in order], name='Sequence'))
    t1 = time.perf_counter()
    if not args.quiet:
        print(f'Collapsing finished in {round(t1-t0, 4)} second(s)\n')
    outlog.write(f'Collapsing finished in {round(t1-t0, 4)} second(s)\n')
    complete_set = complete_set.assign(**dict.fromkeys(PASS_COLUMNS, ''))
    complete_set = complete_set.assign(**dict.fromkeys(['annotFlag'], '0'))
    complete_set = complete_set.reindex(columns=['annotFlag'] + PASS_COLUMNS + list(inFileBaseArray))
    complete_set = complete_set.astype({"annotFlag": int})
    EndTime = time.perf_counter()
    if not args.quiet:
        print(f'Data pre-processing completed in {round(EndTime-begningTime, 4)} second(s)\n')
    outlog.write(f'\nData pre-processing completed in {round(EndTime-begningTime, 4)} second(s)\n\n')
    outlog.close()
    return (complete_set, sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique)
