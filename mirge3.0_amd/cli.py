"""Command line for the hot path only -- the part of ``mirge/__main__.py:24-173`` that is in scope.

  python -m mirge3_amd.cli -s S1.fastq,S2.fastq -lib /path/Libs -on human -db miRBase -o out

Same flag names as the reference (``mirge/libs/parse.py``) for what is implemented -- trimming (-a / -g / -q / -nxt / -NX /
-u), UMIs (-umi / --qiagenumi / -udd), -spk, -tcf, -spl / -rr, -ie, -gff, -ai --; switches of subsystems that are out of
scope (novel miRNA, BAM, tRF, DESeq2, miREC) are rejected instead of being ignored.  Writes the reference's files:
``run.log``, ``mapped.csv``, ``unmapped.csv``, ``miR.Counts.csv``, ``miR.RPM.csv``, ``annotation.report.csv/html``
(+ ``isomirs.csv`` / ``isomirs.samples.csv`` with ``-ie``, ``sample_miRge3.gff`` with ``-gff``, the three
``a2IEditing.*`` files with ``-ai``, ``<sample>_umiCounts.csv`` with ``-udd``).

One process: all samples on one GPU, byte-compatible outputs.  Under ``torch.distributed.run`` with N
ranks: samples are sharded one per GPU (multigpu.py); rank 0 gathers each sample's count columns and its
dictionary with the annotation, merges the dictionaries into the sample matrix on its GPU and writes the same
files as the one-process run (``fastpath.run_sharded_rank0``).
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

DB_KEYS = {"mirbase": "miRBase", "mirgenedb": "MirGeneDB"}  # __main__.py:36


def parse_args(argv=None):
    ap = argparse.ArgumentParser(prog="miRge3.0-amd", description="miRge3.0 hot path on MI355X")
    ap.add_argument("-s", "--samples", required=True, nargs="*",
                    help="comma separated FASTQ files (*.fastq, *.fq, also .gz), a directory of them, or a .txt / .csv file that "
                         "lists one per line (mirge/__main__.py:88-118)")
    ap.add_argument("-o", "--outDir", default=None)
    ap.add_argument("-onam", "-dn", "--outDirName", dest="outDirName", default=None)
    ap.add_argument("-lib", "--libraries-path", dest="libraries_path", required=True)
    ap.add_argument("-on", "--organism-name", dest="organism_name", required=True)
    ap.add_argument("-db", "--mir-DB", dest="mir_DB", default="miRBase")
    ap.add_argument("-ex", "-cr", "--crThreshold", dest="crThreshold", default="0.1")
    ap.add_argument("-m", "--minimum-length", dest="minimum_length", type=int, default=16)
    ap.add_argument("-spk", "--spikeIn", dest="spikeIn", action="store_true")
    ap.add_argument("-shh", "--quiet", action="store_true")
    ap.add_argument("-spl", "--save-pkl", dest="save_pkl", action="store_true",
                    help="save collapsed.pkl / collapsed_accessories.pkl after collapsing (mirge/__main__.py:142-148)")
    ap.add_argument("-rr", "--resume", action="store_true",
                    help="-s names a directory holding those two files: skip the collapse (mirge/__main__.py:91-108)")
    ap.add_argument("-umi", "--uniq-mol-ids", dest="uniq_mol_ids", default=None,
                    help="f,b: drop f bases at the 5' end and b at the 3' end of every read before collapsing")
    ap.add_argument("-udd", "--umiDedup", dest="umiDedup", action="store_true",
                    help="with -umi: count distinct UMI-tagged reads per insert (writes <sample>_umiCounts.csv)")
    ap.add_argument("-tcf", "--tcf-out", dest="tcf_out", action="store_true",
                    help="write <sample>.trim.collapse.fa")
    ap.add_argument("-ie", "--isoform-entropy", dest="isoform_entropy", action="store_true",
                    help="write isomirs.csv and isomirs.samples.csv (isomiR RPMs and entropies)")
    # -a and -g fill ONE list of (kind, sequence) in command-line order, as mirge/libs/parse.py does: the order decides ties
    # between two adapters, and what 'illumina' stands for (mirge/__main__.py:65-83)
    ap.add_argument("-a", "--adapter", dest="adapters", action="append", default=None, type=lambda x: ("back", x),
                    help="3' adapter removed from every read (cutadapt's regular 3' adapter; 'illumina' = TGGAATTCTCGGGTGCCAAGGAACTCCAG)")
    ap.add_argument("-g", "--front", dest="adapters", action="append", default=None, type=lambda x: ("front", x),
                    help="5' adapter: the adapter and everything in front of it are removed (cutadapt's regular 5' adapter; "
                         "'illumina' = GTTCAGAGTTCTACAGTCCGACGATC); with two adapters the better match is removed from a read")
    ap.add_argument("-q", "--quality-cutoff", dest="quality_cutoff", default="10", help="[5'CUTOFF,]3'CUTOFF (default 10, as the reference)")
    ap.add_argument("-nxt", "--nextseq-trim", dest="nextseq_trim", type=int, default=None)
    ap.add_argument("-NX", "--trim-n", dest="trim_n", action="store_true")
    ap.add_argument("-u", "--cut", dest="cut", action="append", type=int, default=[])
    ap.add_argument("--overlap", type=int, default=3)
    ap.add_argument("--error-rate", dest="error_rate", type=float, default=0.12)
    ap.add_argument("-phr", "--phred64", dest="phred64", type=int, default=33)
    ap.add_argument("--trim-count", dest="trim_count", choices=["per-modifier", "once"], default="per-modifier",
                    help="per-modifier: a read is counted after every modifier of the chain, as the reference's worker does "
                         "(digest.py:354-373); once: only the fully trimmed read")
    ap.add_argument("-gff", "--gff-out", dest="gff_out", action="store_true",
                    help="write sample_miRge3.gff (miRTop GFF3 of the miRNA reads, isomiR variant types from the GPU)")
    ap.add_argument("-ai", "--AtoI", dest="AtoI", action="store_true",
                    help="A-to-I editing report (a2IEditing.report.csv, .newform.csv, .detail.txt); the genome filter runs "
                         "`bowtie` against <org>_genome as the reference does (-pbwt / PATH), or reads --genome-retained")
    ap.add_argument("-pbwt", "--bowtie-path", dest="bowtie_path", default=None,
                    help="directory of the bowtie binary used by -ai for the two whole-genome runs")
    ap.add_argument("--genome-retained", dest="genome_retained", default=None,
                    help="-ai without bowtie: file of the miRNA reads with a unique best genome alignment (one per line)")
    ap.add_argument("--genome-aligned", dest="genome_aligned", default=None,
                    help="-ai without bowtie: file of the edited canonical sequences that align to the genome")
    ap.add_argument("-cpu", "--threads", dest="threads", type=int, default=0, help="accepted; only -ai's bowtie runs use it")
    ap.add_argument("--device", type=int, default=None)
    ap.add_argument("--backend", choices=("gpu", "bowtie"), default="gpu",
                    help="gpu (default): the cascade on the MI355X.  bowtie: the reference's ten bowtie runs across its process "
                         "boundary (-pbwt / PATH; collapse and the count tables stay as they are): BASELINE's C1 configuration and a "
                         "parity switch -- both backends must write the same tables")
    # flags of mirge/libs/parse.py that the reference itself never reads (-M, -l, --gc-content, -cms, --compression-level,
    # -op, --numba-*), that only size its worker chunks (--buffer-size), that name tools or carry parameters of the
    # subsystems refused below (-psam, -prf, -mdt, -kh/-ks/-ke, -minl ... -clc), or that it fills in itself (-cuv, -buv):
    # accepted and ignored, so that a reference command line keeps working
    for flags in (("-M", "--maximum-length"), ("-l", "--length"), ("--gc-content",), ("-cms", "--chunkmbs"), ("--compression-level",),
                  ("-op", "--output"), ("--buffer-size",), ("-cuv", "--cutadaptVersion"), ("-buv", "--bowtieVersion"),
                  ("-psam", "--samtools-path"), ("-prf", "--RNAfold-path"), ("-mdt", "--metadata"), ("-kh", "--threshold"),
                  ("-ks", "--kmer-start"), ("-ke", "--kmer-end"), ("-minl", "--minLength"), ("-maxl", "--maxLength"),
                  ("-c", "--minReadCounts"), ("-mloc", "--maxMappingLoci"), ("-sl", "--seedLength"), ("-olc",), ("-clc",)):
        ap.add_argument(*flags, dest="ignored_" + flags[0].strip("-").replace("-", "_"), default=None, help=argparse.SUPPRESS)
    for flag in ("--numba-pll", "--numba-cuda"):
        ap.add_argument(flag, dest="ignored_" + flag.strip("-").replace("-", "_"), action="store_true", help=argparse.SUPPRESS)
    # cutadapt's adapter options: the defaults are what k_trim implements; anything else is refused, not ignored
    ap.add_argument("-n", "--times", dest="times", type=int, default=1, help="remove adapters up to COUNT times from a read (cutadapt -n)")
    ap.add_argument("--no-indels", dest="indels", action="store_false", default=True, help="adapter alignment with substitutions only")
    ap.add_argument("--action", dest="action", default="trim", choices=("trim", "mask", "lowercase", "none"), help=argparse.SUPPRESS)
    ap.add_argument("--match-read-wildcards", dest="match_read_wildcards", action="store_true", default=False, help=argparse.SUPPRESS)
    ap.add_argument("-N", "--no-match-adapter-wildcards", dest="match_adapter_wildcards", action="store_false", default=True, help=argparse.SUPPRESS)
    ap.add_argument("-qumi", "--qiagenumi", dest="qiagenumi", action="store_true",
                    help="with -umi 0,b and -a: the UMI is the b bases that follow the 3' adapter (Qiagen libraries)")
    for flags in (("-nmir", "--novel-miRNA"), ("-bam", "--bam-out"), ("-trf", "--tRNA-frag"), ("-mEC", "--miREC"), ("-dex", "--diffex")):
        ap.add_argument(*flags, dest="oos_" + flags[0].strip("-"), default=None, nargs="?", const=True,
                        help=argparse.SUPPRESS)
    from . import __version__ as _version
    ap.add_argument("--version", action="version", version=str(_version))
    args = ap.parse_args(argv)
    for k, v in vars(args).items():
        if k.startswith("oos_") and v is not None:
            ap.error(f"-{k[4:]} belongs to a miRge3.0 subsystem outside the MI355X hot path (DESIGN.md section 0)")
    if args.action in ("mask", "lowercase") or args.times < 1:
        ap.error("--action mask / lowercase change a read's letters, not its bounds: not part of the MI355X path; -n must be >= 1")
    if (args.umiDedup or args.qiagenumi) and not args.uniq_mol_ids:
        ap.error("-udd / --qiagenumi require -umi f,b")
    if args.qiagenumi and not (args.adapters and args.adapters[0][0] == "back"):
        ap.error("--qiagenumi reads the UMI behind the 3' adapter: give it with -a (first)")
    args.bowtieVersion = "True"
    if (args.AtoI or args.gff_out) and (args.save_pkl or args.resume):
        ap.error("-ai / -gff run on the device-resident route: not together with -spl / -rr")
    if args.backend == "bowtie" and (args.AtoI or args.gff_out or args.isoform_entropy):
        ap.error("--backend bowtie writes the count tables and mapped.csv / unmapped.csv; -ai / -gff / -ie take per-read data of the GPU cascade")
    return args


FASTQ_SUFFIXES = (".fastq", ".fastq.gz", ".fq", ".fq.gz")


def validate_files(args, in_files, runlog, loud=True):
    """``validate_files`` (mirge/libs/miRgeEssential.py:102-127): of the names given, the existing files that end in
    .fastq / .fq (optionally .gz), resolved; a sample's name is its file name without the last extension (the last two
    for .gz).  The others are reported and left out; no file left is an error."""
    full, names = [], []
    with open(runlog, "a+") as outlog:
        for f in in_files:
            p = Path(f)
            # the extension the name ends in (longest first: '.fastq.gz' before '.gz'-less forms); what is in front of it -- it
            # must not be empty -- names the sample
            ext = next((e for e in sorted(FASTQ_SUFFIXES, key=len, reverse=True) if p.name.endswith(e)), None)
            if p.exists() and ext is not None and len(p.name) > len(ext):
                full.append(str(p.resolve()))
                names.append(p.name[:-len(ext)])
                continue
            why = f"\nWARNING: File {f} does not exists!" if not p.exists() else f"\nWARNING: File {f} is neither fastq or fastq.gz format!"
            if loud and not args.quiet:
                print(why)
                print(f"Omitting file {f}")
            if loud:
                outlog.write(why + "\n" + f"Omitting file {f}\n")
        if not full:
            if loud:
                outlog.write("\nERROR!: No valid input files were available!\nPlease verify miRge -s arguments\n")
            sys.exit("\nERROR!: No valid input files were available!\nPlease verify miRge -s arguments\n")
    return full, names


def collect_samples(args, runlog, loud=True):
    """The forms ``-s`` takes (mirge/__main__.py:85-118; only its first value is read, :88): a comma separated list of
    files, a directory (its files, sorted; with -rr the directory of the pickles), or a .txt / .csv file with one name
    per line."""
    file_list = args.samples[0].split(",") if args.samples else []
    if not file_list:
        sys.exit("\nERROR!: No valid input files were available!\nPlease verify miRge -s arguments\n")
    first = Path(file_list[0])
    if first.is_dir():
        if args.resume:
            return file_list, []  # the pickles name the samples
        file_list = sorted(str(x) for x in first.iterdir() if x.is_file())
    elif first.exists() and first.suffix in (".txt", ".csv"):
        with open(first) as fh:
            file_list = [line.strip() for line in fh]
    full, names = validate_files(args, file_list, runlog, loud)
    if loud and not args.quiet:
        print(f"\nmiRge3.0 will process {len(full)} out of {len(file_list)} input file(s).\n")
    if loud:
        with open(runlog, "a+") as outlog:
            outlog.write(f"\nmiRge3.0 will process {len(full)} out of {len(file_list)} input file(s).\n\n")
    return full, names


def main(argv=None):
    globalstart = time.perf_counter()
    args = parse_args(argv)
    # pandas writes the three small tables at the very end; importing it (0.15-0.2 s) runs beside the libraries' load and the
    # GPU work instead of in front of the first table
    import threading
    threading.Thread(target=lambda: __import__("pandas"), daemon=True).start()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MIRGE_SHARE_GPU=1 (test hook): every rank uses device 0 of a single-GPU box
    args.device = (0 if os.environ.get("MIRGE_SHARE_GPU") else local_rank) if args.device is None else args.device
    ref_db = DB_KEYS.get(args.mir_DB.lower()) or sys.exit("ERROR: Require valid database (-d miRBase or MirGeneDB)")
    dist = None
    if world > 1:  # before the run's directory is named: the ranks must agree on it (multigpu.agree_on_run_directory)
        import torch.distributed as dist
        dist.init_process_group("gloo")  # per-sample results of some MB: host-side gather, no device collective
    from . import multigpu
    name = multigpu.agree_on_run_directory(args.outDirName, rank, world, dist)
    workDir = Path(args.outDir or Path.cwd()) / name
    workDir.mkdir(exist_ok=True, parents=True)
    if rank == 0:
        with open(workDir / "run.log", "a+") as fh:
            fh.write(" ".join(sys.argv) + "\n")
    if world > 1 and (args.resume or args.save_pkl):
        sys.exit("-spl / -rr are single-process options")
    if not (Path(args.libraries_path) / args.organism_name / "index.Libs").exists():
        sys.exit("\n ERROR: The path to miRge libraries is incorrect or does not exist!\n")
    if args.organism_name == "hamster":  # mirge/__main__.py:61-64
        if "mirbase" in args.mir_DB.lower() and rank == 0:
            print("Library for hamster is not developed for miRBase, therefore, MirGeneDB is used\n")
        ref_db = "MirGeneDB"
    files, base_names = collect_samples(args, workDir / "run.log", rank == 0)
    from .collapse import unpinned_trim_options
    unpinned = unpinned_trim_options(args)
    if unpinned and rank == 0:  # counted with, but not silently: these follow cutadapt's documentation, no real cutadapt has confirmed them
        msg = ("WARNING: trimming option(s) " + "; ".join(unpinned) + " are implemented from cutadapt's documentation and have not been "
               "compared with cutadapt itself (tools/cutadapt_crosscheck.py does that where cutadapt is installed)")
        with open(workDir / "run.log", "a+") as fh:
            fh.write(msg + "\n")
        if not args.quiet:
            print(msg)

    from . import fastpath
    from .cascade import bwt_align
    from .collapse import baking
    from .countjoin import summarize, finish_tables
    if world == 1:
        import pickle
        if args.resume:  # same two files, same order of their contents as the reference writes them
            import pandas as pd
            rootToPKL = Path(files[0]).absolute()
            if not (rootToPKL / "collapsed.pkl").exists() or not (rootToPKL / "collapsed_accessories.pkl").exists():
                sys.exit("\nERROR: The provided path doesn't contain pickle files with .pkl extensions!\n")
            df = pd.read_pickle(rootToPKL / "collapsed.pkl")
            with open(rootToPKL / "collapsed_accessories.pkl", "rb") as pklin:
                src, trimmed, uniq, files, base_names = pickle.load(pklin)
        elif fastpath.eligible(args) and args.backend == "gpu":
            # everything between the files' text and the count tables stays on the GPU (fastpath.py); the three
            # reference-signature functions below remain the drop-ins for the reference's own call sites
            fastpath.run(args, files, base_names, workDir, ref_db)
            if rank == 0 and not args.quiet:
                print(f"\nThe analysis completed in {round(time.perf_counter() - globalstart, 4)} second(s)\n")
            return
        else:
            df, src, trimmed, uniq = baking(args, files, base_names, str(workDir))
        if args.save_pkl and not args.resume:
            df.to_pickle(workDir / "collapsed.pkl")
            with open(workDir / "collapsed_accessories.pkl", "wb") as pklac:
                pickle.dump([src, trimmed, uniq, files, base_names], pklac, protocol=pickle.HIGHEST_PROTOCOL)
        if args.backend == "bowtie":
            from .cascade import bwt_align_bowtie
            df = bwt_align_bowtie(args, df, str(workDir), ref_db)
        else:
            df = bwt_align(args, df, str(workDir), ref_db)
        pdMapped, pdUnmapped = df[df.annotFlag.eq(1)], df[df.annotFlag.eq(0)]
        summarize(args, str(workDir), ref_db, base_names, pdMapped, src, trimmed, uniq)
        pdMapped.to_csv(workDir / "mapped.csv")
        pdUnmapped.to_csv(workDir / "unmapped.csv")
    else:
        if args.backend == "bowtie":
            sys.exit("--backend bowtie is a single-process option")
        from .cascade import get_cascade
        casc = get_cascade(args, ref_db, args.device)

        if not fastpath.eligible(args):
            sys.exit("-spl / -rr are single-process options")
        via_files = multigpu.directory_is_shared(workDir, rank, world, dist)  # False on a node that does not see rank 0's directory

        # round 6: the run's ONE mapped.csv / unmapped.csv written range by range by EVERY rank (fastpath.run_sharded_ranges) when the
        # run allows it and every rank sees the run's directory; else rank 0 builds the joint table alone (round 5's route)
        use_ranges = via_files and fastpath.parallel_tail_eligible(args, len(files), world, casc)
        held = {} if use_ranges else None

        def process(i):  # device-resident per sample (fastpath.run_sample_tables)
            return fastpath.run_sample_tables(args, files[i], base_names[i], i, workDir, ref_db, casc, via_files,
                                              dictionary_order=len(files) == 1, hold=held)

        t_shard = time.perf_counter()
        tables = multigpu.run_sharded(len(files), rank, world, process, dist)
        t_gather = time.perf_counter()
        tm_ranges = None
        if use_ranges:
            tm_ranges = fastpath.run_sharded_ranges(args, held, len(files), base_names, workDir, casc, rank, world, dist)
            every_tm = [None] * world if rank == 0 else None
            dist.gather_object({k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm_ranges.items()}, every_tm, dst=0)
        if rank == 0:  # the same files as the one-process run: count tables, ONE mapped.csv / unmapped.csv, -gff / -ai / -ie
            tm = {}
            if use_ranges:
                t_ct = time.perf_counter()
                fastpath.sharded_count_tables(args, tables, workDir, ref_db, casc, tm)
                tm["ranges_tail_per_rank"] = every_tm
                tm["rank0_tail_s"] = tm_ranges["range_tail_s"] + (time.perf_counter() - t_ct)
                tm["range_tail_s_slowest_rank"] = max(x["range_tail_s"] for x in every_tm)
            else:
                fastpath.run_sharded_rank0(args, tables, workDir, ref_db, casc, timings=tm)
            # where a sharded run's time goes: every rank's sample, then rank 0's serial tail
            import json
            import resource
            line = {"ranks": world, "samples": len(files), "samples_and_gather_s": round(t_gather - t_shard, 4),
                    "per_sample": [dict(t.timing or {}, name=t.name, index=t.index) for t in tables],
                    "tail": "ranges (every rank its stretch of mapped.csv / unmapped.csv)" if use_ranges else "rank 0 alone",
                    "rank0_tail": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm.items() if not isinstance(v, (list, dict))},
                    "ranges_tail_per_rank": tm.get("ranges_tail_per_rank"),
                    "rank0_peak_rss_MB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)}
            with open(workDir / "run.log", "a+") as fh:
                fh.write("sharded run timing: " + json.dumps(line) + "\n")
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not args.quiet:
        print(f"\nThe analysis completed in {round(time.perf_counter() - globalstart, 4)} second(s)\n")


if __name__ == "__main__":
    main()
