"""The single-process CLI run without a per-read Python object anywhere.

``cli.main`` used to chain the three reference-signature functions (``baking`` -> ``bwt_align`` -> ``summarize``): a
4 M-row DataFrame of Python strings was built, turned back into packed reads for the cascade, and its name columns
were looked up again for the join.  Those functions stay what they are -- drop-ins for the reference's call sites
(``mirge/__main__.py:140,157,166``) -- but the CLI itself now keeps everything between the FASTQ text and the count
tables on the GPU:

    file text --mirge_reads_parse--> raw reads --mirge_collapse_cascade / mirge_collapse + mirge_cascade_run-->
    unique reads + annotation --mirge_count_join--> per-class / per-miRNA tables --finish_tables--> the three CSVs
    unique reads + annotation + counts --one fetch each--> mirge_annotation_csv --> mapped.csv / unmapped.csv

Same files, same bytes (tests/test_gpu_parity.py replays the golden cases through it).
"""
from __future__ import annotations

import time
from pathlib import Path
from typing import Dict, List

import numpy as np

from . import PASS_COLUMNS, _ffi
from .cascade import PASSES, get_cascade
from .collapse import parse_sample, read_text, read_texts, trim_from_args, umi_from_args, write_tcf
from .countjoin import summarize_device
from .seqio import FlatSeqs, load_merges


def eligible(args) -> bool:
    """flags the device-resident run covers (-umi / --qiagenumi / -udd / -tcf included); -spl / -rr take the DataFrame route:
    their pickles ARE the frame"""
    return not (getattr(args, "save_pkl", False) or getattr(args, "resume", False))


def row_order(seqs: FlatSeqs, first: np.ndarray, n_samples: int) -> np.ndarray:
    """Row order of the reference's DataFrame: dict insertion order = first appearance for one sample
    (digest.py:158-163), the sorted union of the sequences for several (pandas ``join(how='outer')``, digest.py:243)."""
    if n_samples == 1:
        # `first` holds distinct raw-read indices: ranking them is one scatter and one compress, not a sort
        if len(first) == 0:
            return np.zeros(0, dtype=np.int64)
        slot = np.full(int(first.max()) + 1, -1, dtype=np.int32)
        slot[first] = np.arange(len(first), dtype=np.int32)
        return slot[slot >= 0].astype(np.int64)
    ln = seqs.lengths
    width = int(ln.max()) if len(seqs) else 1
    mat = np.zeros((len(seqs), width), dtype=np.uint8)  # NUL-padded: a prefix sorts before its extensions, as str does
    rows = np.repeat(np.arange(len(seqs), dtype=np.int64), ln)
    mat[rows, np.arange(int(seqs.offsets[-1]), dtype=np.int64) - seqs.offsets[:-1][rows]] = seqs.data
    return np.argsort(mat.view(f"S{width}").reshape(-1), kind="stable")


def names_by_pass(casc) -> list:
    """Reference names of every pass's library as flat arrays (what ``mirge_annotation_csv`` prints from); built once per
    cascade -- a human library set holds ~0.2 M names, and joining them costs as much as formatting a sample's table."""
    cached = getattr(casc, "_names_by_pass", None)
    if cached is None:
        cached = [FlatSeqs.from_list(casc.libs[PASSES[p][1]].names) if PASSES[p][1] in casc.libs else None
                  for p in range(casc.n_pass)]
        casc._names_by_pass = cached
    return cached


def run_sample_tables(args, file: str, name: str, index: int, workDir, ref_db: str, casc=None, via_files: bool = True):
    """One sample on this process's GPU, for the sharded CLI (one sample per rank, multigpu.py): device-resident parse ->
    collapse + cascade -> count join.  Returns the sample's ``SampleTables`` (a few kB: its columns of the count tables)
    with its ``SampleReads`` attached (unique reads in dictionary order, counts, annotation: what rank 0 needs for the
    run's ONE mapped.csv / unmapped.csv and the per-read reports, ~15 B per unique read)."""
    from . import multigpu
    from .cascade import EXACT_PASS, ISO_PASS
    workDir = Path(workDir)
    casc = casc or get_cascade(args, ref_db, getattr(args, "device", 0))
    ctx = casc.ctx
    raw, n_rec = parse_sample(ctx, read_text(str(file), stream=True), int(getattr(args, "minimum_length", 16)), trim_from_args(args),
                              umi_from_args(args), workDir, name)
    n_trimmed = len(raw)
    if getattr(args, "tcf_out", False):
        write_tcf(workDir / (str(name) + ".trim.collapse.fa"), raw)
    iupac = raw.iupac_seen
    uniq, res = casc.collapse_and_run(raw)
    raw.close()
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, EXACT_PASS, ISO_PASS, len(casc.libs["mirna"]))
    counts, _ = uniq.counts()
    order = uniq.first_appearance_order()
    seqs = uniq.unpack().take(order)
    ps, ref, _, _ = res.fetch()
    out = multigpu.SampleTables(index, name, n_rec, n_trimmed, len(uniq), cls[:, 0], ex[:, 0], iso[:, 0])
    reads = multigpu.SampleReads(seqs.data, seqs.offsets, counts[order, 0], ps[order], ref[order], iupac)
    # handed to rank 0 through files in the run's directory when rank 0 sees that directory (same node / shared filesystem:
    # a sample's dictionary is tens to hundreds of MB), in-band with the tables otherwise
    out.reads = reads.to_files(workDir / ".mirge_shards", index) if via_files else reads
    res.close(); uniq.close()
    return out


def merge_sample_reads(ctx: _ffi.Context, parts):
    """The per-sample dictionaries of a sharded run (``multigpu.SampleReads``, in sample order) -> the run's sample matrix:
    one weighted collapse on this GPU with a sample id per entry (``mirge_collapse_weighted``; the outer join of
    digest.py:243 without expanding the dictionaries again).  -> (uniq DeviceReads with the U x S counts, pass[U], ref[U])
    -- a read's annotation depends on its sequence alone, so the first sample that holds it supplies it."""
    S = len(parts)
    data = np.concatenate([p.data for p in parts]) if S else np.zeros(0, np.uint8)
    lens = np.concatenate([np.diff(p.offsets) for p in parts]) if S else np.zeros(0, np.int64)
    off = np.zeros(lens.shape[0] + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    sid = np.repeat(np.arange(S, dtype=np.int32), [len(p.counts) for p in parts])
    w = np.concatenate([p.counts for p in parts]).astype(np.uint32) if S else np.zeros(0, np.uint32)
    raw = _ffi.DeviceReads.pack(ctx, FlatSeqs(data, off))
    uniq = raw.collapse(sid if S > 1 else None, max(S, 1), weights=w)
    raw.close()
    _, first = uniq.counts()
    ps_all = np.concatenate([p.ps for p in parts]) if S else np.zeros(0, np.int8)
    ref_all = np.concatenate([p.ref for p in parts]) if S else np.zeros(0, np.int32)
    return uniq, ps_all[first], ref_all[first]


def run(args, files: List[str], base_names: List[str], workDir, ref_db: str, timings: Dict[str, float] = None):
    """FASTQ files -> every output file of the hot path.  Returns the dict of ``finish_tables``."""
    t0 = time.perf_counter()
    tm = timings if timings is not None else {}
    workDir = Path(workDir)
    outlog = open(workDir / "run.log", "a+")

    def say(msg):
        if not args.quiet:
            print(msg)
        outlog.write(msg + "\n")

    casc = get_cascade(args, ref_db, getattr(args, "device", 0))
    ctx = casc.ctx
    tm["libraries_s"] = time.perf_counter() - t0
    if tm["libraries_s"] > 0.005:  # this call loaded them: where the time went
        tm["libraries_detail_s"] = {k: round(v, 4) for k, v in casc.timing.items()}
    min_len = int(getattr(args, "minimum_length", 16))
    trim = trim_from_args(args)
    umi = umi_from_args(args)
    sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique = {}, {}, {}
    parsed = []
    t_read = t_parse = 0.0
    from . import collapse as _collapse
    del _collapse.GZ_LOG[:]
    texts = read_texts(files, stream=True)  # read ahead on worker threads; a .gz is inflated on all cores, or piece by piece beside its parse
    gz_tm: Dict[str, float] = {}
    for f, name in zip(files, base_names):
        t = time.perf_counter()
        text = next(texts)
        t_read += time.perf_counter() - t
        t1 = time.perf_counter()
        raw, n_rec = parse_sample(ctx, text, min_len, trim, umi, workDir, name, timings=gz_tm)
        del text
        t_parse += time.perf_counter() - t1
        sampleReadCounts[name], trimmedReadCounts[name] = n_rec, len(raw)
        if getattr(args, "tcf_out", False):
            write_tcf(workDir / (str(name) + ".trim.collapse.fa"), raw)
        if raw.iupac_seen:
            say(f"WARNING: {name} holds IUPAC ambiguity codes other than N (or '.'); they are aligned -- and printed -- as N")
        parsed.append(raw)
        say(f'Cutadapt finished for file {name} in {round(time.perf_counter() - t, 4)} second(s)')
    tm["read_files_s"], tm["parse_s"] = t_read, t_parse
    if _collapse.GZ_LOG:  # .gz input inflated on all host cores (mirge_gz_inflate): inside read_files_s
        tm["gz_parallel"] = list(_collapse.GZ_LOG)
    if gz_tm:  # streamed .gz input: parse_s is then bounded by the inflation (inflate_s, on its worker thread), of which
        tm["gz_stream"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in gz_tm.items()}  # upload_parse_s is the GPU side's share
    t = time.perf_counter()
    say("Alignment in progress ...")
    S = len(parsed)
    # the probe tables for these libraries and read lengths: built once per process (0.05-0.08 s of device work for the human
    # set), nothing afterwards -- its own stage, so that `collapse_cascade_s` is the sample's work
    for p in parsed:
        if len(p):
            casc.prepare(p)
    tm["probe_tables_s"] = time.perf_counter() - t
    t = time.perf_counter()
    if S == 1:
        uniq, res = casc.collapse_and_run(parsed[0])
    else:
        allr = _ffi.DeviceReads.concat(ctx, parsed)
        sid = np.repeat(np.arange(S, dtype=np.int32), [len(p) for p in parsed])
        uniq = allr.collapse(sid, S)
        allr.close()
        res = casc.run(uniq)
    for p in parsed:
        p.close()
    counts = first = None  # fetched by `reports` only if something on the host needs the matrix
    if S == 1:  # every unique read of the one sample has a count
        trimmedReadCountsUnique[base_names[0]] = len(uniq)
    else:
        for name, nz in zip(base_names, uniq.nonzero_per_sample()):
            trimmedReadCountsUnique[name] = int(nz)
    ctx.sync()
    tm["collapse_cascade_s"] = time.perf_counter() - t
    say(f'Alignment completed in {round(time.perf_counter() - t, 4)} second(s)\n')
    t = time.perf_counter()
    merges = load_merges(str(args.libraries_path), args.organism_name, ref_db)
    out = summarize_device(ctx, uniq, res, casc.libs["mirna"], merges,
                           list(base_names), sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique,
                           float(args.crThreshold), bool(args.spikeIn), workDir)
    tm["join_tables_s"] = time.perf_counter() - t
    outlog.close()
    out = reports(args, workDir, ref_db, base_names, casc, uniq, res, out, merges, counts, first, tm)
    tm["total_s"] = time.perf_counter() - t0
    return out


def reports(args, workDir, ref_db: str, base_names, casc, uniq, res, out, merges, counts, first, tm=None, ann=None):
    """What follows the count tables, from the run's joint table (unique reads x samples): ``mapped.csv`` /
    ``unmapped.csv`` (mirge/__main__.py:164-173) and the per-read miRNA reports (-gff, -ai, -ie).  ``ann`` = (pass, ref) per
    unique read when the annotation came from elsewhere (the sharded run gathers it from the ranks); ``res`` may then be
    None unless -gff / -ai ask for the device kernels."""
    tm = tm if tm is not None else {}
    workDir = Path(workDir)
    ctx = casc.ctx
    S = len(base_names)
    # ---- the per-read tables (mirge/__main__.py:164-173)
    t = time.perf_counter()
    want_reports = any(getattr(args, k, False) for k in ("gff_out", "AtoI", "isoform_entropy"))
    # annotation on the device, no per-read report asked for: the two files are formatted on the GPU and neither the reads
    # nor the counts nor the annotation are fetched (they are 35 B per unique read, the files' text 48 B)
    on_device = ann is None and res is not None and not getattr(args, "host_csv", False)
    seqs = ps = ref = off = mm = None
    if counts is None and (not on_device or want_reports):
        counts, first = uniq.counts()
    if not on_device or want_reports:
        seqs = uniq.unpack()
        if ann is None:
            ps, ref, off, mm = res.fetch()
        else:
            ps, ref = ann
    tm["fetch_reads_annotation_s"] = time.perf_counter() - t
    # row order of the reference's frame: dictionary order for one sample (digest.py:158-163), the sorted union of the
    # sequences for several (pandas' outer join, digest.py:243) -- both from a sort on the device
    order = uniq.first_appearance_order() if S == 1 else uniq.sorted_order()
    tm["row_order_s"] = time.perf_counter() - t - tm["fetch_reads_annotation_s"]
    n_cols = 10 if args.spikeIn else 9  # bwtAlign drops the 'spike-in' column when -spk is off (manifoldAlign.py:137-138)
    cols = PASS_COLUMNS[:n_cols]
    header = ",".join(["Sequence", "annotFlag"] + cols + list(base_names)) + "\n"
    done = False
    if on_device:
        done = _ffi.annotation_csv_device(ctx, uniq, res, workDir / "mapped.csv", workDir / "unmapped.csv", header, order,
                                          list(range(casc.n_pass)), n_cols, names_by_pass(casc))
    if not done:
        if seqs is None:
            seqs = uniq.unpack()
            ps, ref, off, mm = res.fetch()
        if counts is None:
            counts, first = uniq.counts()
        _ffi.annotation_csv(workDir / "mapped.csv", workDir / "unmapped.csv", header, seqs, ps, ref, counts, order,
                            list(range(casc.n_pass)), n_cols, names_by_pass(casc))
    tm["per_read_csv_s"] = time.perf_counter() - t
    if getattr(args, "gff_out", False):  # -gff (summary.py:800-837)
        from .gff import write_gff
        t = time.perf_counter()
        out["gff"] = write_gff(args, workDir, ref_db, base_names, casc, uniq, res, seqs, ps, ref, counts, order)
        tm["gff_s"] = time.perf_counter() - t
    if getattr(args, "AtoI", False):  # -ai (summary.py:1034-1057)
        from .a2i import ListedGenome, a2i_report
        t = time.perf_counter()
        genome = getattr(args, "genome_predicate", None)
        if genome is None and getattr(args, "genome_retained", None):
            genome = ListedGenome.from_files(args.genome_retained, getattr(args, "genome_aligned", None))
        out["a2i"] = a2i_report(args, workDir, ref_db, base_names, casc, uniq, res, seqs, ps, ref, counts, order, out, merges,
                                genome=genome)
        tm["a2i_report_s"] = time.perf_counter() - t
    if getattr(args, "isoform_entropy", False):  # -ie reads the miRNA rows of the mapped frame: build just those
        from .countjoin import isomir_entropy_tables
        isomir_entropy_tables(mirna_frame(seqs, ps, ref, counts, order, casc, base_names), base_names, out["filtered"], workDir)
    out["device"] = dict(ctx=ctx, casc=casc, uniq=uniq, res=res, seqs=seqs, ann=(ps, ref, off, mm), counts=counts, order=order)
    return out


def run_sharded_rank0(args, tables, workDir, ref_db: str, casc):
    """Rank 0 of the sharded CLI, after the gather: the count tables from the ranks' own per-sample columns, then the
    run's joint table (``merge_sample_reads``) and everything ``reports`` writes from it -- the same files, byte for byte,
    as the one-process run of the same samples."""
    from . import multigpu
    from .countjoin import finish_tables
    workDir = Path(workDir)
    names, src, trimmed, uniq_n, cls, ex, iso = multigpu.merge_tables(tables)
    merges = load_merges(str(args.libraries_path), args.organism_name, ref_db)
    out = finish_tables(cls, ex, iso, casc.libs["mirna"], merges, names, src, trimmed, uniq_n, float(args.crThreshold),
                        bool(args.spikeIn), workDir=workDir)
    for t in tables:
        if isinstance(t.reads, str):
            t.reads = multigpu.SampleReads.from_files(t.reads)
    try:
        (workDir / ".mirge_shards").rmdir()
    except OSError:
        pass
    with open(workDir / "run.log", "a+") as outlog:
        for t in tables:
            if t.reads.iupac:
                outlog.write(f"WARNING: {t.name} holds IUPAC ambiguity codes other than N (or '.'); they are aligned -- and printed -- as N\n")
    uniq, ps, ref = merge_sample_reads(casc.ctx, [t.reads for t in tables])
    counts, first = uniq.counts()
    res = None
    if getattr(args, "gff_out", False) or getattr(args, "AtoI", False):
        # the two report kernels read the annotation on the device: once more over the joint table (milliseconds), which
        # also cross-checks what the ranks sent
        res = casc.run(uniq)
        ps2, ref2, _, _ = res.fetch()
        if not (np.array_equal(ps2, ps) and np.array_equal(ref2[ps2 >= 0], ref[ps >= 0])):
            raise RuntimeError("sharded run: the annotation gathered from the ranks differs from rank 0's own")
    return reports(args, workDir, ref_db, names, casc, uniq, res, out, merges, counts, first, ann=(ps, ref))


def mirna_frame(seqs: FlatSeqs, ps, ref, counts, order, casc, base_names):
    """the rows of the mapped frame that carry an 'exact miRNA' or 'isomiR miRNA' name, in frame order, as the small
    DataFrame the host-side miRNA reports (-ie, -gff, -ai) iterate over"""
    import pandas as pd
    from .cascade import EXACT_PASS, ISO_PASS
    sel = order[(ps[order] == EXACT_PASS) | (ps[order] == ISO_PASS)]
    names = np.asarray(casc.libs["mirna"].names, dtype=object)
    sub = seqs.take(sel)
    df = pd.DataFrame(counts[sel].astype(np.int64), columns=list(base_names), index=pd.Index(sub.to_list(), name="Sequence"))
    nm = names[ref[sel]] if sel.size else np.zeros(0, dtype=object)
    df.insert(0, "isomiR miRNA", np.where(ps[sel] == ISO_PASS, nm, ""))
    df.insert(0, "exact miRNA", np.where(ps[sel] == EXACT_PASS, nm, ""))
    return df
