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


def run_sample_tables(args, file: str, name: str, index: int, workDir, ref_db: str, casc=None, via_files: bool = True,
                      dictionary_order: bool = False, hold: dict = None):
    """One sample on this process's GPU, for the sharded CLI (one sample per rank, multigpu.py): device-resident parse ->
    collapse + cascade -> count join.  Returns the sample's ``SampleTables`` (a few kB: its columns of the count tables)
    with its ``SampleReads`` attached (unique reads and counts: what rank 0 needs for the run's ONE mapped.csv /
    unmapped.csv and the per-read reports, ~30 B per unique read; in dictionary order only when asked)."""
    from . import multigpu
    from .cascade import EXACT_PASS, ISO_PASS
    workDir = Path(workDir)
    t_sample = time.perf_counter()
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
    t_tail = time.perf_counter()
    if hold is not None:
        # the parallel tail (run_sharded_ranges): the dictionary stays on the device until the run's range splitters are known
        out = multigpu.SampleTables(index, name, n_rec, n_trimmed, len(uniq), cls[:, 0], ex[:, 0], iso[:, 0], iupac=bool(iupac))
        res.close()
        hold[index] = uniq
        out.timing = {"sample_s": round(time.perf_counter() - t_sample, 4), "unique_reads": int(len(uniq))}
        return out
    counts, _ = uniq.counts()
    seqs = uniq.unpack()
    cnt = counts[:, 0]
    if dictionary_order:
        # the order of first appearance only matters when the run has ONE sample (the frame of several is the sorted union of
        # their sequences, digest.py:243): at C4's size the host-side reordering of 7.8 M ragged reads was 2.4 of a rank's 2.9 s
        order = uniq.first_appearance_order()
        seqs, cnt = seqs.take(order), cnt[order]
    out = multigpu.SampleTables(index, name, n_rec, n_trimmed, len(uniq), cls[:, 0], ex[:, 0], iso[:, 0], iupac=bool(iupac))
    # (round 5: no annotation travels -- it depends on the sequence alone, and rank 0 annotates the run's joint table on its own
    # GPU in milliseconds, which puts that table's mapped.csv / unmapped.csv on the device-formatted route; lengths as bytes
    # or 16-bit words instead of 64-bit offsets: 30 B per unique read instead of 47)
    reads = multigpu.SampleReads.from_seqs(seqs, cnt, iupac)
    # handed to rank 0 through files in the run's directory when rank 0 sees that directory (same node / shared filesystem:
    # a sample's dictionary is tens to hundreds of MB), in-band with the tables otherwise
    out.reads = reads.to_files(workDir / ".mirge_shards", index) if via_files else reads
    res.close(); uniq.close()
    out.timing = {"sample_s": round(time.perf_counter() - t_sample, 4), "handover_s": round(time.perf_counter() - t_tail, 4),
                  "unique_reads": int(len(reads.counts))}
    return out


def merge_sample_reads(ctx: _ffi.Context, parts):
    """The per-sample dictionaries of a sharded run (``multigpu.SampleReads``, in sample order) -> the run's sample matrix:
    one weighted collapse on this GPU with a sample id per entry (``mirge_collapse_weighted``; the outer join of
    digest.py:243 without expanding the dictionaries again).  Every dictionary is packed on the device by itself and the
    packed sets are appended there (``mirge_reads_concat``): no host-side concatenation of the sequences (round 4 joined 1.7 GB
    of ASCII and as many bytes of 64-bit offsets on the host at C4's size).  -> uniq DeviceReads with the U x S counts."""
    S = len(parts)
    if not S:
        raw = _ffi.DeviceReads.pack(ctx, FlatSeqs(np.zeros(0, np.uint8), np.zeros(1, np.int64)))
        uniq = raw.collapse()
        raw.close()
        return uniq
    packed = [_ffi.DeviceReads.pack(ctx, FlatSeqs(p.data, p.offsets)) for p in parts]
    raw = packed[0] if S == 1 else _ffi.DeviceReads.concat(ctx, packed)
    sid = np.repeat(np.arange(S, dtype=np.int32), [len(p.counts) for p in parts])
    w = np.concatenate([p.counts for p in parts]).astype(np.uint32)
    uniq = raw.collapse(sid if S > 1 else None, S, weights=w)
    if S > 1:
        raw.close()
    for p in packed:
        p.close()
    return uniq


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
    else:  # (round 6: every sample collapsed by itself, the dictionaries merged on the device -- collapse.collapse_parsed_samples)
        from .collapse import collapse_parsed_samples
        uniq = collapse_parsed_samples(ctx, parsed)
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
    # (-gff alone needs nothing on the host since round 6: its rows are chosen, typed and formatted on the device)
    from . import gff as _gff
    gff_on_device = bool(getattr(args, "gff_out", False)) and ann is None and res is not None and _gff.device_route()
    want_reports = any(getattr(args, k, False) for k in ("AtoI", "isoform_entropy")) or (bool(getattr(args, "gff_out", False)) and not gff_on_device)
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
        from .gff import write_gff, write_gff_device
        t = time.perf_counter()
        if gff_on_device:
            out["gff"] = write_gff_device(args, workDir, ref_db, base_names, casc, uniq, res, order)
        else:
            out["gff"] = write_gff(args, workDir, ref_db, base_names, casc, uniq, res, seqs, ps, ref, counts, order)
        tm["gff_s"] = time.perf_counter() - t
        tm["gff_stages_s"] = out["gff"].get("timing", {})
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


def sharded_count_tables(args, tables, workDir, ref_db: str, casc, tm: Dict[str, float]):
    """Rank 0 of the sharded CLI: the run's count tables (miR.Counts.csv, miR.RPM.csv, annotation.report.*) from the ranks' own
    per-sample columns (a few kB each) -- summary.py:686-798,882-901 on R x S tables; -> (finish_tables' dict, names, merges)"""
    from . import multigpu
    from .countjoin import finish_tables
    t0 = time.perf_counter()
    names, src, trimmed, uniq_n, cls, ex, iso = multigpu.merge_tables(tables)
    merges = load_merges(str(args.libraries_path), args.organism_name, ref_db)
    out = finish_tables(cls, ex, iso, casc.libs["mirna"], merges, names, src, trimmed, uniq_n, float(args.crThreshold),
                        bool(args.spikeIn), workDir=Path(workDir))
    with open(Path(workDir) / "run.log", "a+") as outlog:
        for tb in tables:
            if tb.iupac:
                outlog.write(f"WARNING: {tb.name} holds IUPAC ambiguity codes other than N (or '.'); they are aligned -- and printed -- as N\n")
    tm["count_tables_s"] = time.perf_counter() - t0
    return out, names, merges


def run_sharded_rank0(args, tables, workDir, ref_db: str, casc, timings: Dict[str, float] = None):
    """Rank 0 of the sharded CLI, after the gather: the count tables from the ranks' own per-sample columns, then the
    run's joint table (``merge_sample_reads``) and everything ``reports`` writes from it -- the same files, byte for byte,
    as the one-process run of the same samples.  Round 5: the joint table is annotated HERE, by one cascade over it on rank
    0's GPU (a read's annotation depends on its sequence alone; 67 M unique reads of eight 20 M-read samples take ~15 ms),
    so nothing per read but the dictionaries travels, and ``mapped.csv`` / ``unmapped.csv`` of the union are formatted on the
    GPU like a one-process run's (``reports``: no fetch of reads, counts or annotation, no formatting on host threads)."""
    from . import multigpu
    tm = timings if timings is not None else {}
    t0 = time.perf_counter()
    workDir = Path(workDir)
    out, names, merges = sharded_count_tables(args, tables, workDir, ref_db, casc, tm)
    t = time.perf_counter()
    for tb in tables:
        if isinstance(tb.reads, str):
            tb.reads = multigpu.SampleReads.from_files(tb.reads)
    try:
        (workDir / ".mirge_shards").rmdir()
    except OSError:
        pass
    tm["load_dictionaries_s"] = time.perf_counter() - t
    t = time.perf_counter()
    uniq = merge_sample_reads(casc.ctx, [tb.reads for tb in tables])
    for tb in tables:
        tb.reads = None  # (hundreds of MB per sample at C4's size)
    casc.ctx.sync()
    tm["merge_sample_reads_s"] = time.perf_counter() - t
    t = time.perf_counter()
    res = casc.run(uniq)
    casc.ctx.sync()
    tm["annotate_joint_table_s"] = time.perf_counter() - t
    tm["joint_unique_reads"] = len(uniq)
    t = time.perf_counter()
    out = reports(args, workDir, ref_db, names, casc, uniq, res, out, merges, None, None, tm=tm)
    tm["reports_s"] = time.perf_counter() - t
    tm["rank0_tail_s"] = time.perf_counter() - t0
    return out


def names_need_quoting(casc) -> bool:
    """a reference name holds a comma, a quote or a line break: mapped.csv then takes the host formatter (mirge_annotation_csv)"""
    for fs in names_by_pass(casc):
        if fs is not None and len(fs) and np.isin(fs.data, np.frombuffer(b',"\n\r', dtype=np.uint8)).any():
            return True
    return False


def parallel_tail_eligible(args, n_samples: int, world: int, casc) -> bool:
    """the sharded run's files can be written range by range on every rank: several samples (one sample's frame is in dictionary
    order, not sorted: digest.py:158-163), no per-read report that needs the joint table in one place (-gff / -ai / -ie), names that
    print as they are.  MIRGE_SHARD_TAIL=rank0 keeps round 5's route (tests compare the two)."""
    import os
    if os.environ.get("MIRGE_SHARD_TAIL", "ranges") != "ranges":
        return False
    if world < 2 or n_samples < 2 or getattr(args, "host_csv", False):
        return False
    if any(getattr(args, k, False) for k in ("gff_out", "AtoI", "isoform_entropy")):
        return False
    return not names_need_quoting(casc)


def run_sharded_ranges(args, held: dict, n_samples: int, base_names, workDir, casc, rank: int, world: int, dist, shard_dir=None,
                       timings: Dict[str, float] = None):
    """EVERY rank of the sharded CLI, after its own samples (their dictionaries still on the device, ``held`` = {sample index:
    DeviceReads}): the run's ONE mapped.csv / unmapped.csv written range by range (multigpu.py, 'The parallel tail').
      1. every sample's k quantile keys -> all ranks (a few kB over gloo) -> the same world - 1 splitters everywhere
      2. every held dictionary split by owner range on the device, each stretch handed to its owner as files
      3. this rank's range: the samples' stretches merged (weighted collapse, S columns), annotated by one cascade, ordered by
         the device sort, its rows' bytes counted; the ranks' byte counts -> every rank's offsets in the two files
      4. rank 0 creates the files (header, final size); every rank formats its rows on the GPU and pwrites them at its offset
    Same bytes as the one-process run of the same samples (tests/test_gpu_parity.py, tools/sharded_c4.py)."""
    import os
    from . import multigpu
    tm = timings if timings is not None else {}
    t0 = time.perf_counter()
    workDir = Path(workDir)
    shard_dir = Path(shard_dir) if shard_dir is not None else workDir / ".mirge_shards"
    ctx = casc.ctx
    # ---- 1. splitters
    t = time.perf_counter()
    mine = [(int(i), int(len(u)), u.range_sample(multigpu.RANGE_SAMPLE_KEYS)) for i, u in sorted(held.items())]
    pool = [None] * world
    dist.all_gather_object(pool, mine)
    flat = sorted((x for part in pool for x in part), key=lambda x: x[0])
    splitters = multigpu.choose_splitters([(u, k) for _, u, k in flat], world)
    tm["splitters_s"] = time.perf_counter() - t
    # ---- 2. split + hand over
    t = time.perf_counter()
    for i, u in sorted(held.items()):
        seqs, cnt, bounds = u.range_split(splitters)
        multigpu.write_parts(shard_dir, i, seqs, cnt, bounds)
        u.close()
    held.clear()
    tm["split_handover_s"] = time.perf_counter() - t
    t = time.perf_counter()
    dist.barrier()
    tm["wait_for_all_parts_s"] = time.perf_counter() - t
    # ---- 3. this rank's range of the joint table
    t = time.perf_counter()
    parts = [multigpu.read_part(shard_dir, i, rank) for i in range(n_samples)]
    tm["load_parts_s"] = time.perf_counter() - t
    t = time.perf_counter()
    uniq = merge_sample_reads(ctx, parts)
    del parts
    ctx.sync()
    tm["merge_sample_reads_s"] = time.perf_counter() - t
    t = time.perf_counter()
    res = casc.run(uniq)
    ctx.sync()
    tm["annotate_joint_table_s"] = time.perf_counter() - t
    tm["joint_unique_reads_of_range"] = len(uniq)
    t = time.perf_counter()
    order = uniq.sorted_order()
    tm["row_order_s"] = time.perf_counter() - t
    t = time.perf_counter()
    n_cols = 10 if args.spikeIn else 9
    cols = PASS_COLUMNS[:n_cols]
    header = (",".join(["Sequence", "annotFlag"] + cols + list(base_names)) + "\n").encode()
    names = names_by_pass(casc)
    passes = list(range(casc.n_pass))
    sizes = _ffi.annotation_csv_device_sizes(ctx, uniq, res, order, passes, n_cols, names)
    if sizes is None:
        raise RuntimeError("a reference name needs CSV quoting (parallel_tail_eligible should have excluded this run)")
    every = [None] * world
    dist.all_gather_object(every, (int(sizes[0]), int(sizes[1]), len(uniq)))
    off_m = len(header) + sum(x[0] for x in every[:rank])
    off_u = len(header) + sum(x[1] for x in every[:rank])
    mapped, unmapped = workDir / "mapped.csv", workDir / "unmapped.csv"
    if rank == 0:
        for path, total in ((mapped, sum(x[0] for x in every)), (unmapped, sum(x[1] for x in every))):
            with open(path, "wb") as fh:
                fh.write(header)
                fh.truncate(len(header) + total)
    dist.barrier()  # the files exist at their final size
    tm["sizes_offsets_s"] = time.perf_counter() - t
    t = time.perf_counter()
    _ffi.annotation_csv_device_at(ctx, uniq, res, mapped, unmapped, off_m, off_u, order, passes, n_cols, names)
    tm["format_write_s"] = time.perf_counter() - t
    res.close(); uniq.close()
    if rank == 0:
        try:
            shard_dir.rmdir()
        except OSError:
            pass
    tm["joint_unique_reads"] = int(sum(x[2] for x in every))
    tm["range_tail_s"] = time.perf_counter() - t0
    return tm


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
