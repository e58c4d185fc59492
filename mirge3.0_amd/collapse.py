"""Read collapse and the sample matrix -- host mirror of ``mirge/libs/digest.py:105-302``.

``baking(args, inFileArray, inFileBaseArray, workDir)`` keeps the reference's signature and
return value ``(DataFrame, sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique)``.
The per-chunk dict counting (``cutadapt()``, digest.py:320-375), the dict merge (:141-163) and
the pandas outer join of the samples (:237-245) are ONE ``mirge_collapse`` call on the GPU: all
samples' reads go in together with a sample id and come back as the U distinct sequences plus a
U x S count matrix.

Scope: reads are taken as already trimmed (adapter/quality trimming is
cutadapt's, third-party and upstream of the path).  Asking for adapter trimming raises instead of
silently skipping it.  Only the length filter of digest.py:348,368 (``--minimum-length``) applies.

UMI handling (digest.py:164-205,305-315,358-365): ``-umi f,b`` slices f bases off
the front and b off the back of every read before the collapse (counts add up); with ``-udd`` the
full reads are collapsed first (on the GPU), ``<sample>_umiCounts.csv`` is written from that result,
and the inserts of the DISTINCT tagged reads are collapsed again, so a count is a number of molecules.
``-qumi`` needs the adapter match of cutadapt and is refused.  ``-tcf`` writes
``<sample>.trim.collapse.fa`` (digest.py:219-229).
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


def trim_from_args(args):
    """The cutadapt modifier chain of ``stipulate`` (digest.py:59-101) as the options of ``mirge_reads_parse_trim``:
    ``-q`` (default "10": quality trimming is ALWAYS in the reference's chain), ``-a`` (one 3' adapter; 'illumina' =
    the TruSeq small-RNA adapter) or ``-g`` (one 5' adapter), ``-nxt``, ``-NX``, ``-u``, ``--overlap``, ``--error-rate``,
    ``-phr``.  Several adapters, ``-n > 1`` and ``--action`` other than trim are refused."""
    adapters = getattr(args, "adapters", None) or []
    if isinstance(adapters, str):
        adapters = [("back", adapters)]
    adapters = [(a if isinstance(a, (tuple, list)) else ("back", a)) for a in adapters]
    front = getattr(args, "front", None) or []
    if isinstance(front, str):
        front = [front]
    adapters += [(f if isinstance(f, (tuple, list)) else ("front", f)) for f in front]
    if len(adapters) > 1 or any(kind not in ("back", "front") for kind, _ in adapters):
        raise NotImplementedError("one adapter is supported: one 3' adapter (-a) or one 5' adapter (-g), not several")
    is_front = bool(adapters) and adapters[0][0] == "front"
    adapter = adapters[0][1] if adapters else None
    if adapter == "illumina":
        adapter = ILLUMINA_3P
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
    return _ffi.MirgeTrim.make(adapter=adapter, quality_back=qb, quality_front=qf, nextseq=-1 if nxt is None else int(nxt),
                               phred_base=base, min_overlap=int(getattr(args, "overlap", 3)),
                               error_rate=float(getattr(args, "error_rate", 0.12)), trim_n=bool(getattr(args, "trim_n", False)),
                               cut=cut, count_per_modifier=getattr(args, "trim_count", "per-modifier") != "once", front=is_front)


def filter_min_length(reads: FlatSeqs, min_len: int) -> FlatSeqs:
    keep = np.flatnonzero(reads.lengths >= int(min_len))
    if keep.shape[0] == len(reads):
        return reads
    return reads.take(keep)


def collapse_samples(ctx: _ffi.Context, samples: Sequence[FlatSeqs]):
    """-> (DeviceReads uniq (with counts), order) where ``order`` lists the unique rows in the
    reference's row order: first appearance for one sample (dict order, digest.py:158-163),
    lexicographic for several (pandas ``join(how='outer')`` sorts, digest.py:243)."""
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


def _collapse_one(ctx: _ffi.Context, reads: FlatSeqs):
    """One sample alone -> (sequences, counts) in dict order (first appearance, digest.py:158-163)."""
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq = raw.collapse(None, 1)
    raw.close()
    counts, first = uniq.counts()
    seqs = uniq.unpack()
    uniq.close()
    order = np.argsort(first, kind="stable")
    return seqs.take(order), counts[order, 0].astype(np.int64)


def baking(args, inFileArray, inFileBaseArray, workDir, ctx: _ffi.Context = None):
    """Drop-in for ``baking`` (digest.py:105-302) on already-trimmed reads."""
    import pandas as pd
    if getattr(args, "qiagenumi", None):
        raise NotImplementedError("-qumi reads cutadapt's adapter match object and is not part of the MI355X path")
    trim = trim_from_args(args)
    umi = getattr(args, "uniq_mol_ids", None)
    if umi and (trim.adapter_len or trim.trim_n or trim.n_cut or trim.nextseq_cutoff >= 0):
        raise NotImplementedError("-umi together with adapter / N / unconditional trimming is not supported: trim first")
    dedup = bool(getattr(args, "umiDedup", False))
    if umi:
        umi_f, umi_b = (int(x) for x in str(umi).split(","))
    begningTime = time.perf_counter()
    runlogFile = Path(workDir) / "run.log"
    outlog = open(str(runlogFile), "a+")
    ctx = ctx or _ffi.Context(getattr(args, "device", 0))
    min_len = int(getattr(args, "minimum_length", 16))
    sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique = {}, {}, {}
    samples: List[FlatSeqs] = []
    # No UMI: every file's text goes to the GPU as it is (records found, filtered by length and packed there:
    # digest.py:320-375 without a per-read host step); the samples' read sets are appended on the device and
    # collapsed together with a sample id (digest.py:141-163 + the outer join :243).  With UMIs the sequences are
    # cut out and sliced on the host first.
    device_parse = not umi
    uniq = None
    parsed: List[_ffi.DeviceReads] = []
    texts = read_texts(inFileArray) if device_parse else None  # read (and gunzipped) ahead on worker threads
    for FQfile, name in zip(inFileArray, inFileBaseArray):
        start = time.perf_counter()
        if device_parse:
            raw, n_rec = _ffi.DeviceReads.parse(ctx, next(texts), 0, min_len, trim)
            sampleReadCounts[name] = n_rec
            trimmedReadCounts[name] = len(raw)
            parsed.append(raw)
            finish2 = time.perf_counter()
            if not args.quiet:
                print(f'Cutadapt finished for file {name} in {round(finish2-start, 4)} second(s)')
            outlog.write(f'Cutadapt finished for file {name} in {round(finish2-start, 4)} second(s)\n')
            if getattr(args, "tcf_out", False):
                u1 = raw.collapse()
                tc, tfirst = u1.counts()
                tl = u1.unpack().to_list()
                u1.close()
                by = sorted(range(len(tl)), key=lambda i: (-int(tc[i, 0]), int(tfirst[i])))  # by count, ties in dict order
                with open(Path(workDir) / (str(name) + '.trim.collapse.fa'), 'w') as fo:
                    fo.write("".join(f">seq{k + 1}_{int(tc[i, 0])}\n{tl[i]}\n" for k, i in enumerate(by)))
            continue
        reads = read_fastq_sequences(str(FQfile))
        sampleReadCounts[name] = len(reads)
        if not umi:
            reads = filter_min_length(reads, min_len)
        else:
            pure, _ = reads.umi_split(umi_f, umi_b)
            keep = np.flatnonzero(pure.lengths >= min_len)  # the worker's filter is on the insert (:360)
            if not dedup:
                reads = pure.take(keep)  # counts add up: slicing first == slicing the collapsed dict (:168-181)
            else:
                full, c = _collapse_one(ctx, reads.take(keep))
                pure, tag = full.umi_split(umi_f, umi_b)
                with open(Path(workDir) / (name + "_umiCounts.csv"), "a+") as iumiFile:  # (:183-197)
                    iumiFile.write("UMISeq,transcriptSeq,UMICounts\n")
                    iumiFile.write("".join(f"{t},{p},{n}\n" for t, p, n in zip(tag.to_list(), pure.to_list(), c.tolist())))
                reads = pure  # one entry per distinct tagged read, in dict order
        trimmedReadCounts[name] = len(reads)
        samples.append(reads)
        finish2 = time.perf_counter()
        if not args.quiet:
            print(f'Cutadapt finished for file {name} in {round(finish2-start, 4)} second(s)')
        outlog.write(f'Cutadapt finished for file {name} in {round(finish2-start, 4)} second(s)\n')
        if getattr(args, "tcf_out", False):  # (:219-229): by count, ties in dict order
            tseqs, tc = _collapse_one(ctx, reads)
            by = np.argsort(-tc, kind="stable")
            tl = tseqs.to_list()
            with open(Path(workDir) / (str(name) + '.trim.collapse.fa'), 'w') as fo:
                fo.write("".join(f">seq{k + 1}_{int(tc[i])}\n{tl[i]}\n" for k, i in enumerate(by)))
    t0 = time.perf_counter()
    if device_parse:
        if len(parsed) == 1:
            uniq = parsed[0].collapse()
        else:
            allr = _ffi.DeviceReads.concat(ctx, parsed)
            sid = np.repeat(np.arange(len(parsed), dtype=np.int32), [len(p) for p in parsed])
            uniq = allr.collapse(sid, len(parsed))
            allr.close()
        for p in parsed:
            p.close()
    else:
        uniq = collapse_samples(ctx, samples)
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
