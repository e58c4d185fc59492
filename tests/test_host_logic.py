"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, the
product fails loudly without a GPU, the count-join tail reproduces the reference's CSVs from
per-miRNA tables, FASTQ parsing, and the oracle's two matchers agree."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from helpers import CASES, GoldenCase, PASS_COLS, oracle_libs_from
import mirge3_amd  # noqa: F401
from mirge3_amd import _ffi, synth
from mirge3_amd.cascade import PASSES, policies
from mirge3_amd.collapse import read_fastq_sequences, filter_min_length
from mirge3_amd.countjoin import finish_tables
from mirge3_amd.seqio import FlatSeqs

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_abi_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    hdr = open(os.path.join(ROOT, "include", "mirge_native.h")).read()
    declared = set(re.findall(r"\b(mirge_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    lib = C.CDLL(_ffi.SO_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_no_kernel_is_built_for_threadgroup_split_mode(monkeypatch):
    """`k_cascade_bulk` reads its workgroup's own survivor lists back with plain loads (kernels_cascade.hpp,
    MIRGE_SURV_PLAIN_LOADS): right only while a workgroup's waves share one CU's L1.  The built code object must not ask for
    tgsplit in any kernel descriptor (compute_pgm_rsrc3 bit 16; `_codeobj` was checked against a -mtgsplit build of a toy
    kernel), and build() refuses the flag."""
    import __graft_entry__ as g
    from mirge3_amd import _codeobj
    g.build()
    assert _codeobj.n_kernels(_ffi.SO_PATH) > 100
    assert _codeobj.tg_split_kernels(_ffi.SO_PATH) == []
    monkeypatch.setenv("HIPCC_COMPILE_FLAGS_APPEND", "-mtgsplit")
    with pytest.raises(RuntimeError, match="tgsplit"):
        g.build()
    monkeypatch.setenv("HIPCC_COMPILE_FLAGS_APPEND", "-mno-tgsplit")
    g.build()


def test_bench_names_the_kernels_the_library_holds():
    """bench.py finds the dominant kernel's PMC rows by a fragment of its demangled name (rocprof_symbol); a changed template
    signature would turn `roofline.traffic` into null without anything failing -- every fragment must name a kernel of the
    built library (its host-side launch stubs carry the names)."""
    import importlib.util
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    so = os.path.join(os.path.dirname(_ffi.__file__), "csrc", "libmirge_native.so")
    names = subprocess.run([nm, "-C", so], capture_output=True, text=True, check=True).stdout
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)  # main() is behind __name__ == "__main__"
    for rec in ("k_cascade_bulk.w1", "k_cascade_bulk.w2", "k_cascade_fused.w2", "k_cascade_fused.w1n", "k_pass[0].w1",
                "k_pass[4-6].w1", "k_pass[8].w2", "k_part_agg.w1", "k_part_dedup.w1", "k_part_split.w1", "k_resolve.w1",
                "k_collapse_insert.w2", "k_collapse_scatter.w2n", "k_join", "k_join_reduce"):
        frag = bench.rocprof_symbol(rec)
        assert frag in names, (rec, frag)


def test_policy_tables_agree_with_oracle():
    # the product's table and the oracle's are written independently from manifoldAlign.py:84-85
    for p, (col, key, args, kw) in enumerate(PASSES):
        ocol, okey, oargs, okw = oracle.PASSES[p]
        assert (col, key) == (ocol, okey)
        assert args.replace("--threads", "").split() == oargs.split()
        for f in ("mode", "mm", "seedlen", "maxtotal", "trim5", "trim3", "ttail", "len_lt", "len_gt"):
            assert kw.get(f, 0) == okw.get(f, 0), (p, f)
    assert C.sizeof(_ffi.MirgePolicy) == 40 and len(policies(10)) == 10


@pytest.mark.skipif(_ffi.load().mirge_device_count() > 0, reason="GPU present")
def test_product_fails_loudly_without_gpu():
    with pytest.raises(RuntimeError, match="no HIP device"):
        _ffi.Context(0)
    from mirge3_amd.cascade import Cascade
    sl = synth.make_libraries(scale="tiny")
    with pytest.raises(RuntimeError):
        Cascade(_ffi.Context(0), sl.libs)


@pytest.mark.parametrize("name", CASES)
def test_count_join_tail_reproduces_reference_csvs(name, tmp_path):
    """finish_tables (the pandas tail behind mirge_count_join) fed with sums computed in numpy
    from the reference's mapped.csv must write the reference's three CSVs byte for byte."""
    import csv
    case = GoldenCase(name)
    exp = case.expected_annotation()
    mir = case.libs["mirna"]
    lut = {}
    for i, nm in enumerate(mir.names):
        lut.setdefault(nm, i)
    S = len(case.samples)
    cls = np.zeros((case.n_pass, S), dtype=np.int64)
    ex = np.zeros((len(mir), S), dtype=np.int64)
    iso = np.zeros((len(mir), S), dtype=np.int64)
    for s, row in zip(case.seqs, case.counts):
        p, nm = exp[s]
        if p < 0:
            continue
        cls[p] += row
        if p == 0:
            ex[lut[nm]] += row
        if p == 8:
            iso[lut[nm]] += row
    finish_tables(cls, ex, iso, mir, case.merges, case.samples, case.sample_read_counts, case.trimmed,
                  case.trimmed_unique, float(case.cr), case.spike, workDir=tmp_path)
    for f in ("annotation.report.csv", "annotation.report.html", "miR.Counts.csv", "miR.RPM.csv"):
        assert (tmp_path / f).read_text() == case.text(f), f


@pytest.mark.parametrize("name", ["case1_single", "case3_spikein", "case7_two_samples_cr0.4"])
def test_bowtie_backend_replays_the_reference_loop(name, tmp_path):
    """`--backend bowtie` (cascade.bwt_align_bowtie): the reference's ten bowtie runs across its process boundary -- subset
    rules, FASTA named by sequence, T-tail heads, argv strings verbatim, SAM consumption (manifoldAlign.py:12-146) -- needs no
    GPU.  Against the stand-in bowtie that made the golden files (tests/golden/fake_bowtie answers from the oracle's matcher):
    the frame it returns must print as the mapped.csv / unmapped.csv the reference's own bwtAlign wrote."""
    import pandas as pd
    from types import SimpleNamespace
    from mirge3_amd.cascade import bwt_align_bowtie
    case = GoldenCase(name)
    df = pd.DataFrame(case.counts, columns=case.samples, index=pd.Index(case.seqs, name="Sequence"))
    df = df.assign(**dict.fromkeys(PASS_COLS, '')).assign(annotFlag=0).reindex(columns=['annotFlag'] + PASS_COLS + case.samples)
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fake_bowtie")
    args = SimpleNamespace(quiet=True, bowtie_path=fake, threads=2, libraries_path=case.libdir, organism_name="human", spikeIn=case.spike)
    out = bwt_align_bowtie(args, df, str(tmp_path), "miRBase")
    out[out.annotFlag.eq(1)].to_csv(tmp_path / "mapped.csv")
    out[out.annotFlag.eq(0)].to_csv(tmp_path / "unmapped.csv")
    for f in ("mapped.csv", "unmapped.csv"):
        assert (tmp_path / f).read_text() == case.text(f), f
    assert "bowtie backend" in (tmp_path / "run.log").read_text() and not (tmp_path / "bwtInput.fasta").exists()


def test_fastq_parsing(tmp_path):
    p = tmp_path / "x.fastq"
    p.write_text("@r1\nACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIII\n@r2\nACGT\n+\nIIII\n@r3\nTTTTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n")
    fs = read_fastq_sequences(str(p))
    assert fs.to_list() == ["ACGTACGTACGTACGTAC", "ACGT", "TTTTACGTACGTACGTACGT"]
    assert filter_min_length(fs, 16).to_list() == ["ACGTACGTACGTACGTAC", "TTTTACGTACGTACGTACGT"]
    import gzip
    with gzip.open(tmp_path / "y.fastq.gz", "wb") as fh:
        fh.write(p.read_bytes())
    assert read_fastq_sequences(str(tmp_path / "y.fastq.gz")).to_list() == fs.to_list()
    (tmp_path / "e.fastq").write_text("")
    assert len(read_fastq_sequences(str(tmp_path / "e.fastq"))) == 0


def test_oracle_bruteforce_vs_indexed_ci_scale():
    sl = synth.make_libraries(seed=21, scale="ci")
    reads = synth.make_reads(sl, 4000, seed=2, n_frac=0.02)
    libs = oracle_libs_from(sl.libs)
    a = oracle.cascade(reads.data, reads.offsets, libs, n_pass=9, indexed=False)
    b = oracle.cascade(reads.data, reads.offsets, libs, n_pass=9, indexed=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # planted truth: most reads drawn from a library are annotated
    assert (a[0] >= 0).mean() > 0.8


def test_synthetic_generator_is_seeded():
    a = synth.make_libraries(seed=3, scale="tiny")
    b = synth.make_libraries(seed=3, scale="tiny")
    for k in a.libs:
        assert np.array_equal(a.libs[k].seqs.data, b.libs[k].seqs.data)
    ra = synth.make_reads(a, 500, seed=4)
    rb = synth.make_reads(b, 500, seed=4)
    assert np.array_equal(ra.data, rb.data) and np.array_equal(ra.offsets, rb.offsets)
    assert ra.lengths.min() >= 16 and len(set(ra.lengths.tolist())) > 3


def test_cli_flags_in_and_out_of_scope():
    """Flag names of mirge/libs/parse.py: hot-path flags parse, other subsystems' flags are refused."""
    from mirge3_amd.cli import parse_args
    base = ["-s", "a.fq", "-lib", "/x", "-on", "human"]
    a = parse_args(base + ["-umi", "4,4", "-udd", "-tcf", "-spk", "-m", "18", "-ie"])
    assert a.uniq_mol_ids == "4,4" and a.umiDedup and a.tcf_out and a.spikeIn and a.minimum_length == 18 and a.isoform_entropy
    assert a.adapters is None and a.qiagenumi is False and a.quality_cutoff == "10"
    q = parse_args(base + ["-a", "AACTGTAGGCACCATCAAT", "--qiagenumi", "-umi", "0,12", "-udd"])  # docs/source/quick_start.md:286
    assert q.qiagenumi and q.umiDedup and q.uniq_mol_ids == "0,12"
    b = parse_args(base + ["-ai", "-pbwt", "/opt/bowtie", "--genome-retained", "r.txt"])
    assert b.AtoI and b.bowtie_path == "/opt/bowtie" and b.genome_retained == "r.txt"
    t = parse_args(base + ["-a", "illumina", "-q", "5,20", "-NX", "-u", "2", "-u", "-1", "--trim-count", "once"])
    from mirge3_amd.collapse import trim_from_args, ILLUMINA_3P
    tr = trim_from_args(t)
    assert tr.adapter == ILLUMINA_3P.encode() and (tr.quality_front, tr.quality_back) == (5, 20) and tr.trim_n == 1
    assert tr.n_cut == 2 and list(tr.cut) == [2, -1] and tr.count_per_modifier == 0 and tr.error_rate == 0.12 and tr.min_overlap == 3
    d = trim_from_args(parse_args(base))  # the reference's chain always holds the quality trimmer (-q default "10")
    assert d.adapter_len == 0 and d.quality_back == 10 and d.count_per_modifier == 1
    with pytest.raises(NotImplementedError):
        trim_from_args(parse_args(base + ["-a", "AAAA", "-a", "CCCC", "-g", "GGGG"]))
    g = trim_from_args(parse_args(base + ["-g", "GTTCAGAGTTCTACAGTCCGACGATC", "--overlap", "5"]))  # one 5' adapter
    assert g.adapter_front == 1 and g.adapter == b"GTTCAGAGTTCTACAGTCCGACGATC" and g.min_overlap == 5 and tr.adapter_front == 0
    # 'illumina' (mirge/__main__.py:65-83): one adapter -- by its kind; two -- the first is the 3' one, the second the 5' one
    from mirge3_amd.collapse import ILLUMINA_5P, adapters_from_args
    assert trim_from_args(parse_args(base + ["-g", "illumina"])).adapter == ILLUMINA_5P.encode()
    two = trim_from_args(parse_args(base + ["-a", "illumina", "-g", "illumina"]))
    assert (two.adapter, two.adapter_front, two.adapter2, two.adapter2_front) == (ILLUMINA_3P.encode(), 0, ILLUMINA_5P.encode(), 1)
    swapped = parse_args(base + ["-g", "illumina", "-a", "illumina"])  # the reference's rule goes by position, not by flag
    assert adapters_from_args(swapped) == [("front", ILLUMINA_3P), ("back", ILLUMINA_5P)]
    aa = trim_from_args(parse_args(base + ["-a", "AAAAC", "-a", "CCCCG"]))
    assert (aa.adapter, aa.adapter2, aa.adapter_front, aa.adapter2_front) == (b"AAAAC", b"CCCCG", 0, 0)
    # anchored and linked specifications (cutadapt's parser; docs/source/quick_start.md:213-220 documents the linked one)
    from mirge3_amd.collapse import parse_adapter_spec, unpinned_trim_options
    lk = trim_from_args(parse_args(base + ["-g", "TTAGGC...TGGAATTCTCGGGTGCCAAGGAACTCCAGT"]))
    assert (lk.linked, lk.adapter, lk.adapter_front, lk.adapter_anchored, lk.adapter2, lk.adapter2_front, lk.adapter2_anchored,
            lk.linked_required) == (1, b"TTAGGC", 1, 0, b"TGGAATTCTCGGGTGCCAAGGAACTCCAGT", 0, 0, 3)
    la = trim_from_args(parse_args(base + ["-a", "TTAGGC...TGGAATTC$"]))
    assert (la.linked, la.adapter_anchored, la.adapter2_anchored, la.linked_required) == (1, 1, 1, 1)
    an = trim_from_args(parse_args(base + ["-g", "^TTAGGC", "-a", "TGGAATTC$"]))
    assert (an.linked, an.adapter, an.adapter_anchored, an.adapter2, an.adapter2_anchored) == (0, b"TTAGGC", 1, b"TGGAATTC", 1)
    assert parse_adapter_spec("front", "^ACGT") == dict(kind="front", seq="ACGT", anchored=True)
    assert "a linked adapter (A...B)" in unpinned_trim_options(parse_args(base + ["-g", "TTAGGC...TGGAATTC"]))
    assert "an anchored adapter (^A / A$)" in unpinned_trim_options(parse_args(base + ["-a", "TGGAATTC$"]))
    for bad in (["-a", "XACGT...TTTT"], ["-g", "XACGTACGT"], ["-a", "ACGTACGTX"], ["-a", "name=ACGT"], ["-a", "ACGT;min_overlap=5"],
                ["-a", "file:adapters.fa"], ["-g", "AAAA...CCCC", "-a", "GGGG"], ["-a", "ACGT..."]):
        with pytest.raises(NotImplementedError):
            trim_from_args(parse_args(base + bad))
    for bad in (["-a", "^ACGT"], ["-g", "ACGT$"]):
        with pytest.raises(SystemExit):
            trim_from_args(parse_args(base + bad))
    x = parse_args(base + ["-ex", "0.2", "-onam", "run1"])  # the reference's spellings of these two flags
    assert x.crThreshold == "0.2" and x.outDirName == "run1"
    # flags the reference never reads, or that belong to tools / subsystems outside the path: accepted, ignored
    k = parse_args(base + ["-M", "40", "-l", "3", "--gc-content", "50", "-cms", "256", "--buffer-size", "4000000", "-cuv", "2.7",
                           "-psam", "/opt/samtools", "-minl", "16", "-olc", "14", "--numba-pll", "-n", "1", "--action", "trim"])
    assert k.minimum_length == 16 and k.adapters is None
    r = trim_from_args(parse_args(base + ["-a", "illumina", "-n", "2", "--no-indels"]))  # cutadapt's -n / --no-indels
    assert (r.times, r.no_indels) == (2, 1) and (tr.times, tr.no_indels) == (1, 0)
    w = trim_from_args(parse_args(base + ["-a", "ACGTNNACGT", "--match-read-wildcards", "-N", "--action", "none"]))
    assert (w.match_read_wildcards, w.no_adapter_wildcards, w.action_none) == (1, 1, 1) and (tr.match_read_wildcards, tr.action_none) == (0, 0)
    for bad in (["-udd"], ["-qumi"], ["-bam"], ["-trf"], ["-nmir"], ["-mEC"], ["-ai", "-spl"], ["-n", "0"], ["--action", "mask"],
                ["--action", "lowercase"]):
        with pytest.raises(SystemExit):
            parse_args(base + bad)


def test_umi_split_matches_python_slices():
    """FlatSeqs.umi_split == UMIParser (digest.py:305-315) for ragged, short and empty sequences."""
    rng = np.random.default_rng(3)
    seqs = ["".join("ACGTN"[int(c)] for c in rng.integers(0, 5, size=int(n))) for n in rng.integers(0, 40, size=300)]
    fs = FlatSeqs.from_list(seqs)
    for f, b in ((4, 4), (0, 4), (4, 0), (3, 30), (0, 0), (50, 2), (2, 50)):
        pure, tag = fs.umi_split(f, b)
        exp = [oracle.umi_parser(s, f, b) for s in seqs]
        assert pure.to_list() == [e[0] for e in exp] and tag.to_list() == [e[1] for e in exp]


@pytest.mark.parametrize("name", CASES)
def test_isomir_entropy_tables_equal_the_reference(name, tmp_path):
    """-ie: isomirs.csv / isomirs.samples.csv from the reference's own mapped.csv must come out byte-identical to
    what the reference's create_ie wrote (summary.py:915-1021; tests/golden made with isoform_entropy=True)."""
    import pandas as pd
    from mirge3_amd.countjoin import isomir_entropy_tables
    case = GoldenCase(name)
    mapped = pd.read_csv(os.path.join(case.dir, "mapped.csv"), index_col=0, keep_default_na=False)
    counts = pd.read_csv(os.path.join(case.dir, "miR.Counts.csv"), index_col=0)
    filtered = counts[case.samples].sum(axis=0).to_dict()  # Filtered miRNA Reads (summary.py:766)
    isomir_entropy_tables(mapped, case.samples, filtered, tmp_path)
    for f in ("isomirs.csv", "isomirs.samples.csv"):
        assert (tmp_path / f).read_text() == case.text(f), f


def test_gff_name_resolution_tables(tmp_path):
    """What create_gff looks up per miRNA NAME (summary.py:96-186), for both annotation layouts: miRBase (primary
    transcript lines name the precursor by their last ';' field, mature lines by their third) and MirGeneDB (pre_miRNA
    lines / ID= fields); '.SNP' suffixes and -3p/-5p fallbacks; the last precursor's empty sequence."""
    from mirge3_amd import gff
    from mirge3_amd.seqio import FlatSeqs, Library
    mb = tmp_path / "mb.gff3"
    mb.write_text("##gff-version 3\n"
                  "chr1\t.\tmiRNA_primary_transcript\t100\t180\t.\t+\t.\tID=MI1;Alias=MI1;Name=hsa-mir-1\n"
                  "chr1\t.\tmiRNA\t110\t131\t.\t+\t.\tID=MIMAT1;Alias=MIMAT1;Name=hsa-miR-1-5p;Derives_from=MI1\n"
                  "chr1\t.\tmiRNA\t150\t171\t.\t+\t.\tID=MIMAT2;Alias=MIMAT2;Name=hsa-miR-1;Derives_from=MI1\n"
                  "chr2\t.\tmiRNA_primary_transcript\t500\t580\t.\t-\t.\tID=MI2;Alias=MI2;Name=hsa-mir-2\n"
                  "chr2\t.\tmiRNA\t510\t531\t.\t-\t.\tID=MIMAT3;Alias=MIMAT3;Name=hsa-miR-2-3p;Derives_from=MI2\n"
                  "chr2\t.\tmiRNA\t510\t531\t.\t-\t.\tID=MIMAT9;Alias=MIMAT9;Name=hsa-miR-1-5p;Derives_from=MI2\n")
    pre_of = gff.read_annotation(mb, "miRBase")
    assert pre_of == {"hsa-miR-1-5p": "hsa-mir-1", "hsa-miR-1": "hsa-mir-1", "hsa-miR-2-3p": "hsa-mir-2"}
    mg = tmp_path / "mg.gff3"
    mg.write_text("chr1\t.\tpre_miRNA\t100\t160\t.\t+\t.\tID=Hsa-Mir-1_pre;Alias=MI1\n"
                  "chr1\t.\tmiRNA\t110\t131\t.\t+\t.\tID=Hsa-Mir-1_5p;Alias=MIMAT1\n"
                  "chr1\t.\tmiRNA\t140\t160\t.\t+\t.\tID=Hsa-Mir-1_3p*;Alias=MIMAT2\n")
    assert gff.read_annotation(mg, "MirGeneDB") == {"Hsa-Mir-1_5p": "Hsa-Mir-1_pre", "Hsa-Mir-1_3p*": "Hsa-Mir-1_pre"}
    hp = Library(["hsa-mir-1", "hsa-mir-2"], FlatSeqs.from_list(["GGGGACGTACGTACGTACGTACGGGG", "CCCCTTTTGGGGAAAACCCC"]),
                 ["hsa-mir-1 extra words", "hsa-mir-2"])
    pre = gff.precursor_dict(hp)
    assert pre == {"hsa-mir-1": "GGGGACGTACGTACGTACGTACGGGG", "hsa-mir-2": ""}  # the last one is emptied (summary.py:819-826)
    mirDict = {"hsa-miR-1-5p": "ACGTACGTACGTACGTACG", "hsa-miR-1": "TTTTGGGG", "hsa-miR-2-3p": "TTTTGGGGAAAA"}
    names = ["hsa-miR-1-5p", "hsa-miR-1-5p.SNP3", "hsa-miR-1-3p", "hsa-miR-2-3p", "hsa-miR-7-5p"]
    t = gff.resolve_names(names, mirDict, pre_of, pre)
    assert t["master_of_ref"].tolist() == [0, 0, 1, 2, -1]
    assert [t["printed"][i] for i in t["name_of_ref"][:4]] == ["hsa-miR-1-5p", "hsa-miR-1-5p", "hsa-miR-1", "hsa-miR-2-3p"]
    assert t["start0"].tolist() == [5, 0, 1]  # found at 4 (+1); not in its precursor -> find() = -1 -> 0; empty precursor -> 1
    assert [t["parents"][i] for i in t["pre_of_master"]] == ["hsa-mir-1", "hsa-mir-1", "hsa-mir-2"]
    fa = tmp_path / "m.fa"
    fa.write_text(">a\nACGT\n>b desc\nGG\nTT\n")
    assert gff.read_mature_fasta(fa) == {"a": "ACGT", "b desc": "TT"}


def test_read_texts_keeps_file_order_and_inflates_gz(tmp_path):
    """collapse.read_texts: the files' bytes in the order given, whatever the worker threads' finishing order; .gz inflated,
    wrapped FASTA unwrapped, an empty file an empty text."""
    import gzip
    from mirge3_amd.collapse import read_texts, read_text
    paths, want = [], []
    for k in range(7):
        body = b"".join(b"@r%d\nACGT%s\n+\nIIII%s\n" % (i, b"A" * (k + i % 3), b"I" * (k + i % 3)) for i in range(50 * (7 - k)))
        p = tmp_path / (f"s{k}.fastq.gz" if k % 2 else f"s{k}.fastq")
        if k % 2:
            with gzip.open(p, "wb") as fh:
                fh.write(body)
        else:
            p.write_bytes(body)
        paths.append(p); want.append(body)
    fa = tmp_path / "w.fa"; fa.write_bytes(b">a\nACGT\nACGT\n>b\nTT\n"); paths.append(fa); want.append(b">a\nACGTACGT\n>b\nTT\n")
    empty = tmp_path / "e.fastq"; empty.write_bytes(b""); paths.append(empty); want.append(b"")
    got = [bytes(t) for t in read_texts(paths, depth=3)]
    assert got == want
    assert [bytes(t) for t in read_texts(paths[:1])] == want[:1] and bytes(read_text(str(paths[1]))) == want[1]


def test_gz_inflate_on_all_cores_equals_zlib(tmp_path):
    """mirge_gz_inflate (csrc/native_gz.hpp; host only -- no GPU involved): a sample.fastq.gz of a few MB comes back byte for byte
    as zlib inflates it, through `_ffi.gz_inflate` and through `collapse.read_text`, also a file of several members (lanes merged
    with cat); what the route does not take -- a small file, a damaged one -- comes back as None / goes to the streamed zlib
    route, which reports the damage."""
    import gzip
    import zlib
    from mirge3_amd import _ffi, collapse
    rng = np.random.default_rng(9)
    n = 120000
    L = rng.integers(16, 51, size=n)
    recs = []
    for i in range(n):
        sq = "".join("ACGT"[x] for x in rng.integers(0, 4, int(L[i])))
        ql = "".join(chr(33 + int(x)) for x in rng.integers(2, 41, int(L[i])))
        recs.append(f"@SRR1.{i} {i} length={L[i]}\n{sq}\n+\n{ql}\n")
    text = "".join(recs).encode()
    p1 = tmp_path / "S.fastq.gz"
    p1.write_bytes(gzip.compress(text, 6))
    assert p1.stat().st_size > (3 << 20)
    for threads in (0, 1, 3):
        got = _ffi.gz_inflate(p1.read_bytes(), threads)
        assert got is not None and got.tobytes() == text
    del collapse.GZ_LOG[:]
    rt = collapse.read_text(str(p1))
    assert isinstance(rt, np.ndarray) and rt.tobytes() == text and collapse.GZ_LOG and collapse.GZ_LOG[0]["text_MB"] == round(len(text) / 1e6, 1)
    st = collapse.read_text(str(p1), stream=True)  # a caller that streams gets the text while it inflates
    assert isinstance(st, collapse.ParallelGzipStream) and b"".join(bytes(q) for q in st) == text
    # lanes merged with `cat`: a large member, a small one (zlib's), an empty one, a large one, zero padding
    two = tmp_path / "T.fastq.gz"
    two.write_bytes(gzip.compress(text, 6) + gzip.compress(text[:5000], 6) + gzip.compress(b"", 6) + gzip.compress(text, 9) + b"\0" * 512)
    got = _ffi.gz_inflate(two.read_bytes())
    assert got is not None and got.tobytes() == text + text[:5000] + text
    rt2 = collapse.read_text(str(two))
    assert isinstance(rt2, np.ndarray) and rt2.tobytes() == text + text[:5000] + text
    assert b"".join(bytes(q) for q in collapse.read_text(str(two), stream=True)) == text + text[:5000] + text
    os.environ["MIRGE_GZ_PARALLEL"] = "0"
    try:
        st = collapse.read_text(str(two), stream=True)
        assert isinstance(st, collapse.GzipRecordStream) and b"".join(st) == text + text[:5000] + text
    finally:
        os.environ.pop("MIRGE_GZ_PARALLEL")
    # MIRGE_GZ_PARALLEL=whole with stream=True (how fastpath and bench.py's `inflated_whole_then_parsed` leg call it): the WHOLE
    # file inflated on all cores first, then handed over -- not the streamed zlib route (round 5's advisor found it fell there)
    os.environ["MIRGE_GZ_PARALLEL"] = "whole"
    try:
        del collapse.GZ_LOG[:]
        wt = collapse.read_text(str(p1), stream=True)
        assert isinstance(wt, np.ndarray) and wt.tobytes() == text
        assert collapse.GZ_LOG and "parallel_inflate_s" in collapse.GZ_LOG[-1]
    finally:
        os.environ.pop("MIRGE_GZ_PARALLEL")
    assert _ffi.gz_inflate(p1.read_bytes() + b"junk behind the member") is None
    assert _ffi.gz_inflate(gzip.compress(text[:100000], 6)) is None  # too small to cut
    bad = bytearray(p1.read_bytes())
    bad[len(bad) // 2] ^= 4
    assert _ffi.gz_inflate(bytes(bad)) is None
    (tmp_path / "B.fastq.gz").write_bytes(bytes(bad))
    with pytest.raises(collapse.GzRouteDeclined):  # (_parse_stream then starts over with zlib, which says what is wrong)
        list(collapse.read_text(str(tmp_path / "B.fastq.gz"), stream=True))
    with pytest.raises((zlib.error, EOFError, OSError)):
        list(collapse.GzipRecordStream(str(tmp_path / "B.fastq.gz")))


def test_unpinned_trimming_options_are_named():
    """Options whose cutadapt behaviour is restated without a real cutadapt's vectors behind it are listed for the run log (the
    advisor's finding: accepted options must not change counts silently); the plain chain (-a, -q) is not among them."""
    from mirge3_amd.cli import parse_args
    from mirge3_amd.collapse import unpinned_trim_options
    base = ["-s", "x.fastq", "-lib", "/x", "-on", "human", "-shh"]
    assert unpinned_trim_options(parse_args(base + ["-a", "illumina"])) == []
    got = unpinned_trim_options(parse_args(base + ["-a", "illumina", "-g", "ACGTACGT", "-n", "2", "--no-indels", "--action", "none",
                                                   "--match-read-wildcards", "-N"]))
    assert len(got) == 6 and any("two adapters" in g for g in got) and any("--no-indels" in g for g in got)


def test_text_record_stream(tmp_path, monkeypatch):
    """collapse.TextRecordStream: a FASTQ text in memory (bytes, an array, a memory-mapped file) as pieces of whole 4-line records --
    how a text of 8 GiB or more reaches mirge_reads_parse, which takes less at a time.  Every piece size: the pieces are the text,
    each ends a record; quality lines that start with '@' or '+' do not mislead the cut (records are counted, not guessed); a text
    without a final newline; a record longer than a piece raises; read_text hands such a stream out only for FASTQ, only when the
    caller streams, only beyond the size one call takes."""
    from mirge3_amd import collapse
    rng = np.random.default_rng(3)
    recs = []
    for i in range(20000):
        L = int(rng.integers(16, 60))
        q = "".join(chr(c) for c in rng.integers(33, 75, L))
        if i % 5 == 0:
            q = "@" + q[1:]
        if i % 7 == 0:
            q = "+" + q[1:]
        recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[x] for x in rng.integers(0, 4, L)), q))
    text = "".join(recs).encode()
    starts = set(np.cumsum([0] + [len(r) for r in recs]).tolist())
    plain = tmp_path / "a.fastq"
    plain.write_bytes(text)
    for data, want in ((text, text), (np.frombuffer(text, np.uint8), text), (np.memmap(plain, dtype=np.uint8, mode="r"), text), (text[:-1], text[:-1])):
        for piece in (300, 4096, 70000, 1 << 20, 1 << 30):
            st = collapse.TextRecordStream(data, piece_bytes=piece)
            pieces = [bytes(p_) for p_ in st]
            assert b"".join(pieces) == want and st.text_bytes == len(want) and st.pieces == len(pieces)
            at = 0
            for q in pieces:
                assert at in starts and len(q) <= piece
                at += len(q)
            assert len(pieces) >= min(len(want) // piece, 1) and (piece < (1 << 20) or len(pieces) == (2 if piece == 1 << 20 else 1))
    with pytest.raises(RuntimeError, match="record longer"):
        list(collapse.TextRecordStream(text, piece_bytes=40))
    assert list(collapse.TextRecordStream(b"")) == []
    assert bytes(collapse.TextRecordStream(text).whole_text()) == text
    # read_text: the stream only where it is needed
    monkeypatch.setattr(collapse, "TEXT_PIECE_BYTES", 100000)
    st = collapse.read_text(str(plain), stream=True)
    assert isinstance(st, collapse.TextRecordStream) and b"".join(bytes(p_) for p_ in st) == text and st.pieces > 10
    assert not isinstance(collapse.read_text(str(plain), stream=False), collapse.TextRecordStream)
    fa = tmp_path / "a.fa"
    fa.write_bytes(b"".join(b">s%d\n%s\n" % (i, b"ACGT" * 8) for i in range(20000)))
    assert not isinstance(collapse.read_text(str(fa), stream=True), collapse.TextRecordStream)
    monkeypatch.setattr(collapse, "TEXT_PIECE_BYTES", 2 << 30)
    assert not isinstance(collapse.read_text(str(plain), stream=True), collapse.TextRecordStream)


def test_parallel_gzip_stream_parsed_beside_the_inflation(tmp_path):
    """collapse.ParallelGzipStream (mirge_gz_inflate_progress on its own thread, pure host): the pieces handed out while the file
    inflates are the text, each starts at a record -- quality lines that start with '@' included --, one member or several; a FASTA
    comes whole; a damaged file raises GzRouteDeclined from the iteration (the caller starts over with zlib) and whole_text() falls
    back by itself; collapse.record_cut on hand-made texts."""
    import gzip
    from mirge3_amd import collapse
    rc = collapse.record_cut
    t = np.frombuffer(b"@a\nACGT\n+\n@III\n@b\nAC\n+\nII\n@c\nA", np.uint8)
    assert rc(t, 0, t.size) == 15 and bytes(t[15:17]) == b"@b"   # '@c' has no '+' two lines on yet; '@III' is a quality line
    assert rc(t, 0, 14) == 0 and rc(t, 0, 3) == 0 and rc(t, 15, t.size) == 15
    t2 = np.frombuffer(b"@a\n\n+\n\n@b\n\n+\n\n", np.uint8)             # empty reads
    assert rc(t2, 0, t2.size) == 7
    rng = np.random.default_rng(9)
    recs = []
    for i in range(120000):
        L = int(rng.integers(16, 60))
        q = "".join(chr(c) for c in rng.integers(33, 75, L))
        if i % 3 == 0:
            q = "@" + q[1:]
        recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[x] for x in rng.integers(0, 4, L)), q))
    text = "".join(recs).encode()
    starts = set(np.cumsum([0] + [len(r) for r in recs]).tolist())
    one = tmp_path / "a.fastq.gz"
    one.write_bytes(gzip.compress(text, 6))
    two = tmp_path / "b.fastq.gz"
    c = len(text) // 2 + 5
    two.write_bytes(gzip.compress(text[:c], 6) + gzip.compress(text[c:], 1))
    assert one.stat().st_size > (3 << 20)
    for path in (one, two):
        for piece in (1 << 16, 1 << 20, 48 << 20):
            st = collapse.ParallelGzipStream(str(path), piece_bytes=piece)
            pieces = [bytes(p_) for p_ in st]
            st.close()
            assert b"".join(pieces) == text and st.text_bytes == len(text) and st.pieces == len(pieces) and st.inflate_s > 0
            at = 0
            for q in pieces:
                assert at in starts
                at += len(q)
    assert collapse.GZ_LOG and collapse.GZ_LOG[-1]["parsed_beside"] and collapse.GZ_LOG[-1]["text_MB"] == round(len(text) / 1e6, 1)
    # the text buffer of a closed stream serves the next one (no fresh pages per sample); a text handed out whole is never reused
    from mirge3_amd import _ffi
    assert len(_ffi._gz_kept) == 1
    kept = _ffi._gz_kept[0]
    st = collapse.ParallelGzipStream(str(two))
    assert st.job.out is kept and not _ffi._gz_kept
    assert b"".join(bytes(p_) for p_ in st) == text
    st.close()
    assert _ffi._gz_kept and _ffi._gz_kept[0] is kept
    st = collapse.ParallelGzipStream(str(one))
    whole = st.whole_text()
    st.close()
    assert not _ffi._gz_kept and bytes(whole) == text
    assert bytes(collapse.ParallelGzipStream(str(one)).whole_text()) == text
    st = collapse.read_text(str(one), stream=True)
    assert isinstance(st, collapse.ParallelGzipStream)
    st.close()
    # the whole text in host memory only where the host has it: otherwise the streamed route, whose pieces bound the memory
    assert 0 < collapse._host_memory_available() < (1 << 62)
    import unittest.mock
    with unittest.mock.patch.object(collapse, "_host_memory_available", lambda: 16 * one.stat().st_size - 1):
        st = collapse.read_text(str(one), stream=True)
        assert isinstance(st, collapse.GzipRecordStream) and b"".join(st) == text
    assert bytes(collapse.read_text(str(one), stream=False)) == text
    bad = bytearray(one.read_bytes())
    bad[len(bad) // 2] ^= 0x10
    (tmp_path / "bad.fastq.gz").write_bytes(bytes(bad))
    with pytest.raises(collapse.GzRouteDeclined):
        list(collapse.ParallelGzipStream(str(tmp_path / "bad.fastq.gz")))
    with pytest.raises(Exception):  # ... and zlib then says what is wrong with it
        collapse.ParallelGzipStream(str(tmp_path / "bad.fastq.gz")).whole_text()
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 150000 * 55)].tobytes()
    fa_text = b"".join(b">s%d\n%s\n%s\n" % (i, bases[55 * i:55 * i + 40], bases[55 * i + 40:55 * i + 55]) for i in range(150000))
    fa = tmp_path / "c.fa.gz"
    fa.write_bytes(gzip.compress(fa_text, 6))
    items = list(collapse.ParallelGzipStream(str(fa)))
    assert len(items) == 1 and items[0][0] == "whole" and items[0][1] == fa_text
    assert collapse.ParallelGzipStream(str(fa)).whole_text() == collapse.unwrap_fasta(fa_text)


def test_gzip_record_stream(tmp_path):
    """collapse.GzipRecordStream: a .fastq.gz comes out as pieces that are whole 4-line records, in order, byte for byte the
    file's text -- whatever the piece size, for one member, several members (bgzip / cat), an empty member and zero padding
    behind the last, and a text without a final newline; a truncated file raises (as gzip.open does), a FASTA comes whole
    (and unwrapped), a consumer that gives up early does not leave the worker hanging; read_texts(stream=True) opens the
    streams ahead and keeps the files' order."""
    import gzip
    from mirge3_amd import collapse
    rng = np.random.default_rng(1)
    recs = []
    for i in range(30000):
        L = int(rng.integers(16, 40))
        recs.append("@r%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[x] for x in rng.integers(0, 4, L)), "I" * L))
    text = "".join(recs).encode()
    one = tmp_path / "a.fastq.gz"
    with gzip.open(one, "wb", compresslevel=6) as fh:
        fh.write(text)
    many = tmp_path / "b.fastq.gz"
    c1, c2 = len(text) // 3 + 7, 2 * len(text) // 3 + 1
    many.write_bytes(b"".join(gzip.compress(part, 6) for part in (text[:c1], text[c1:c2], b"", text[c2:])) + b"\0" * 100)
    nonl = tmp_path / "c.fastq.gz"
    with gzip.open(nonl, "wb") as fh:
        fh.write(text[:-1])
    for path, want in ((one, text), (many, text), (nonl, text[:-1])):
        for piece in (1 << 12, 1 << 16, 1 << 20, 8 << 20):
            st = collapse.GzipRecordStream(str(path), piece_bytes=piece)
            pieces = list(st)
            assert b"".join(pieces) == want and st.text_bytes == len(want) and st.pieces == len(pieces)
            assert st.compressed_bytes == path.stat().st_size and st.inflate_s > 0
            for q in pieces[:-1]:
                assert q.count(b"\n") % 4 == 0 and q[:1] == b"@" and q[-1:] == b"\n"
            assert len(pieces) >= min(len(want) // (2 * piece), 3)
    # zero padding BETWEEN members and behind the last that spans several 1 MiB reads of the file (round 4's review: the part of
    # it in the next read went to a fresh decompressor, 'incorrect header check'; gzip.open / xopen read such files)
    padded = tmp_path / "pad.fastq.gz"
    padded.write_bytes(gzip.compress(text[:c1], 6) + b"\0" * ((2 << 20) + 77) + gzip.compress(text[c1:], 6) + b"\0" * ((3 << 20) + 5))
    with gzip.open(padded, "rb") as fh:
        assert fh.read() == text
    assert b"".join(collapse.GzipRecordStream(str(padded), piece_bytes=1 << 18)) == text
    cut = tmp_path / "d.fastq.gz"
    cut.write_bytes(one.read_bytes()[: one.stat().st_size // 2])
    with pytest.raises(EOFError):
        list(collapse.GzipRecordStream(str(cut)))
    with pytest.raises(Exception):
        (tmp_path / "g.fastq.gz").write_bytes(b"this is no gzip file")
        list(collapse.GzipRecordStream(str(tmp_path / "g.fastq.gz")))
    fa = tmp_path / "e.fa.gz"
    with gzip.open(fa, "wb") as fh:
        fh.write(b">a\nACGT\nACGT\n>b\nTTTT\n")
    assert list(collapse.GzipRecordStream(str(fa))) == [("whole", b">a\nACGT\nACGT\n>b\nTTTT\n")]
    assert collapse.GzipRecordStream(str(fa)).whole_text() == b">a\nACGTACGT\n>b\nTTTT\n"
    assert collapse.GzipRecordStream(str(one), piece_bytes=1 << 14).whole_text() == text
    st = collapse.GzipRecordStream(str(one), piece_bytes=1 << 12, depth=2)
    next(iter(st))
    st.close()
    assert not st._thread.is_alive()
    plain = tmp_path / "p.fastq"
    plain.write_bytes(text[: text.index(b"@r100\n")])
    got = list(collapse.read_texts([one, plain, many, nonl], depth=2, stream=True))
    assert isinstance(got[0], collapse.GzipRecordStream) and isinstance(got[2], collapse.GzipRecordStream)
    assert bytes(got[1]) == plain.read_bytes()
    assert [b"".join(g) for g in (got[0], got[2], got[3])] == [text, text, text[:-1]]


def test_parallel_inflations_reserve_their_memory(tmp_path, monkeypatch):
    """Admission of the whole-text inflater (round 4's review, medium): read_texts starts up to four inflations at once; each
    used to test 8 x its size against MemAvailable before any had touched a page, so four could be admitted into twice the
    host's memory.  Now a file's need -- 3 x max(ISIZE, 8 x size): the text plus 2-byte symbols, the real size of a very
    repetitive file -- is RESERVED under a lock, at most GZ_MAX_IN_FLIGHT at a time, and given back by close()."""
    import gzip
    from mirge3_amd import collapse
    rec = b"@r\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n"
    rep = tmp_path / "rep.fastq.gz"          # 343:1-like: ISIZE says what 8 x the size does not
    rep.write_bytes(gzip.compress(rec * 400000, 6))
    assert collapse._gz_host_bytes(str(rep)) == 3 * len(rec) * 400000 > 100 * rep.stat().st_size
    rng = np.random.default_rng(5)
    body = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), 30).astype(np.uint8)), b"I" * 30) for i in range(150000))
    files = []
    for k in range(4):
        f = tmp_path / f"s{k}.fastq.gz"
        f.write_bytes(gzip.compress(body, 1))
        files.append(str(f))
    need = collapse._gz_host_bytes(files[0])
    assert need >= 3 * len(body)
    # a host with room for exactly two such files (half of what is left must cover a file's need)
    monkeypatch.setattr(collapse, "_host_memory_available", lambda: 4 * need - 2)
    assert collapse._gz_reserved == 0 and collapse._gz_in_flight == 0
    a = collapse._gz_reserve(files[0])
    b = collapse._gz_reserve(files[1])
    assert a == need and b == need and collapse._gz_in_flight == 2
    assert collapse._gz_reserve(files[2]) == 0            # the third would have passed the old per-file test
    assert not collapse._gz_fits_in_memory(files[2])
    collapse._gz_release(a)
    monkeypatch.setattr(collapse, "GZ_MAX_IN_FLIGHT", 1)
    assert collapse._gz_reserve(files[2]) == 0            # one in flight is the limit now
    collapse._gz_release(b)
    assert collapse._gz_reserved == 0 and collapse._gz_in_flight == 0
    # four threads at once (read_texts): never more than the limit admitted, every reservation returned, texts intact
    monkeypatch.setattr(collapse, "GZ_MAX_IN_FLIGHT", 2)
    monkeypatch.setattr(collapse, "_host_memory_available", lambda: 1 << 40)
    peak = []
    real = collapse._gz_reserve

    def spy(path):
        r = real(path)
        peak.append(collapse._gz_in_flight)
        return r
    monkeypatch.setattr(collapse, "_gz_reserve", spy)
    kinds = []
    for st in collapse.read_texts(files, depth=4, stream=True):
        kinds.append(type(st).__name__)
        try:
            got = b"".join(bytes(p) for p in st)
        except collapse.GzRouteDeclined:
            got = b"".join(collapse.GzipRecordStream(st.path))
        finally:
            st.close()
        assert got == body
    assert max(peak) <= 2 and "ParallelGzipStream" in kinds
    assert collapse._gz_reserved == 0 and collapse._gz_in_flight == 0


def test_sample_forms_of_the_command_line(tmp_path):
    """``-s`` as the reference reads it (mirge/__main__.py:85-118, miRgeEssential.validate_files :102-127): a comma list, a
    directory (sorted), a .txt / .csv list; only *.fastq / *.fq [.gz] files that exist are kept, a sample is named by its
    file name without the last extension (two for .gz); nothing left is an error."""
    from mirge3_amd.cli import collect_samples, parse_args
    d = tmp_path / "in"
    d.mkdir()
    for n in ("b.trimmed.fastq", "a.fq.gz", "c.fastq.gz", "notes.txt", "lib.fa", "d.fq"):
        (d / n).write_text("")
    base = ["-lib", "/x", "-on", "human"]
    log = tmp_path / "run.log"
    a = parse_args(["-s", str(d)] + base + ["-shh"])
    files, names = collect_samples(a, log)
    assert names == ["a", "b.trimmed", "c", "d"] and files == [str((d / n).resolve()) for n in ("a.fq.gz", "b.trimmed.fastq", "c.fastq.gz", "d.fq")]
    assert "Omitting file" in log.read_text() and "will process 4 out of 6 input file(s)" in log.read_text()
    a = parse_args(["-s", f"{d / 'd.fq'},{d / 'missing.fastq'},{d / 'a.fq.gz'}"] + base + ["-shh"])
    assert collect_samples(a, log)[1] == ["d", "a"]
    for ext in (".txt", ".csv"):
        lst = tmp_path / ("list" + ext)
        lst.write_text(f"{d / 'c.fastq.gz'}\n  {d / 'b.trimmed.fastq'}  \n\n")
        a = parse_args(["-s", str(lst)] + base + ["-shh"])
        assert collect_samples(a, log)[1] == ["c", "b.trimmed"]
    with pytest.raises(SystemExit):
        collect_samples(parse_args(["-s", str(d / "lib.fa")] + base + ["-shh"]), log)
    r = parse_args(["-s", str(d), "-rr"] + base)
    assert collect_samples(r, log) == ([str(d)], [])


def test_validate_files_equals_the_reference_function(tmp_path):
    """tests/golden/validate_files.json was written by miRgeEssential.validate_files itself (make_golden.py::run_validate_case): 24
    file names of every shape -- several dots, a blank, .gz twice, upper-case extensions, a bare '.fastq', suffixes that merely
    contain 'fastq' -- plus missing ones; cli.validate_files keeps the same files and gives the samples the same names."""
    import json
    from types import SimpleNamespace
    from mirge3_amd.cli import validate_files
    with open(os.path.join(os.path.dirname(__file__), "golden", "validate_files.json")) as fh:
        d = json.load(fh)
    for n in d["present"]:
        (tmp_path / n).write_text("")
    full, names = validate_files(SimpleNamespace(quiet=True), [str(tmp_path / n) for n in d["given"]], tmp_path / "run.log")
    assert [os.path.basename(f) for f in full] == d["kept"] and names == d["base_names"] and len(full) == 14
    assert full == [str((tmp_path / n).resolve()) for n in d["kept"]]


def test_library_cache_round_trip_and_invalidation(tmp_path, monkeypatch):
    """libcache: the packed image + names written next to an index come back as the same library (lengths at once, letters
    decoded on demand, kept verbatim for small libraries), and the cache is ignored when the index file changed, when the
    layout version differs, or when MIRGE_LIB_CACHE=0; an unwritable directory sends it to the user's cache directory."""
    from mirge3_amd import libcache
    from mirge3_amd.seqio import Library, load_index, write_fasta
    monkeypatch.setenv("MIRGE_LIB_CACHE", "1")
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "xdg"))
    rng = np.random.default_rng(4)
    seqs = ["".join("ACGTN"[int(c)] for c in rng.choice(5, size=int(n), p=[.245, .245, .245, .245, .02])) for n in rng.integers(1, 300, size=60)]
    seqs[3] = "ACGTRYACGT"  # an IUPAC code: invalid for the engine, kept as a letter for the host-side reports
    lib = Library([f"r{i}" for i in range(len(seqs))], FlatSeqs.from_list(seqs), [f"r{i} chr1 segs:x" for i in range(len(seqs))])
    base = str(tmp_path / "human_mrna")
    write_fasta(base + ".fa", lib)
    # the image mirge_lib_create derives (csrc/mirge_libbuild.hpp), in numpy
    lens = lib.seqs.lengths
    total = int(lens.sum() + len(seqs))
    rs = np.zeros(len(seqs) + 1, dtype=np.uint32)
    rs[1:] = np.cumsum(lens + 1)
    T = np.zeros((total + 31) // 32 + 8, dtype=np.uint64)
    inv = np.full((total + 63) // 64 + 4, np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    valid = 0
    for t, q in enumerate(seqs):
        for i, ch in enumerate(q):
            if ch in "ACGT":
                g = int(rs[t]) + i
                T[g >> 5] |= np.uint64("ACGT".index(ch) << (2 * (g & 31)))
                inv[g >> 6] &= ~np.uint64(1 << (g & 63))
                valid += 1
    packed = {"T": T, "inv": inv, "ref_start": rs, "total": total, "kmax": 9, "valid_positions": valid}
    first = load_index(base)
    assert first.cache_target[0] == base and not hasattr(first, "cache_path")
    path = libcache.save(base, first, packed, first.cache_target[1])
    assert path == base + ".mirge3amd"
    got = load_index(base)
    assert got.cache_path == path and got.names == first.names and got.headers == first.headers
    assert np.array_equal(got.seqs.offsets, first.seqs.offsets)
    assert got.seqs.to_list() == first.seqs.to_list()  # small library: the letters themselves, IUPAC code included
    assert np.array_equal(got.seqs.packed["T"], T) and got.seqs.packed["total"] == total and got.seqs.packed["kmax"] == 9
    dec = libcache.decode(packed, np.asarray(first.seqs.offsets))  # what a large library would decode to
    want = "".join(q if set(q) <= set("ACGTN") else "".join(c if c in "ACGT" else "N" for c in q) for q in seqs)
    assert dec.tobytes().decode() == want
    # a changed index file: the cache no longer applies
    os.utime(base + ".fa", ns=(1, 1))
    assert not hasattr(load_index(base), "cache_path")
    libcache.save(base, first, packed, libcache.source_stamp(base))
    assert load_index(base).cache_path == path
    # a damaged cache (one flipped byte in any array, header intact, size and mtime of the INDEX unchanged): the array no
    # longer matches its content hash, the cache is ignored and the next save replaces it
    meta_len = int.from_bytes(open(path, "rb").read()[16:24], "little")
    import json as _json
    meta = _json.loads(open(path, "rb").read()[24:24 + meta_len])
    body0 = (16 + 8 + meta_len + 63) & ~63
    good = open(path, "rb").read()
    for k in ("T", "inv", "ref_start", "names_data", "seq_offsets"):
        d = meta["arrays"][k]
        at = body0 + d["at"] + (np.dtype(d["dtype"]).itemsize * d["n"]) // 2
        bad = bytearray(good)
        bad[at] ^= 0x10
        with open(path, "wb") as fh:
            fh.write(bad)
        assert not hasattr(load_index(base), "cache_path"), k
    with open(path, "wb") as fh:
        fh.write(good)
    assert load_index(base).cache_path == path
    with open(path, "wb") as fh:
        fh.write(bad)
    reread = load_index(base)  # ... read from the FASTA again, and its save() heals the file
    libcache.save(base, reread, packed, reread.cache_target[1])
    assert load_index(base).cache_path == path
    # both hash algorithms verify what they wrote; one that is not available here makes the cache unverifiable, not trusted
    for algo in ("xxh3_64", "blake2b8"):
        h = libcache.content_hash(T, algo)
        assert h is None or (h[0] == algo and libcache.content_hash(T.copy(), algo) == h and libcache.content_hash(T[::-1].copy(), algo) != h)
    assert libcache.content_hash(T, "blake2b8") is not None and libcache.content_hash(T, "no-such-hash") is None
    # a bare index name (no directory part) must not take save() down
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert libcache.save("human_mrna", first, packed, libcache.source_stamp("human_mrna")) == "human_mrna.mirge3amd"
    finally:
        os.chdir(cwd)
    monkeypatch.setattr(libcache, "VERSION", libcache.VERSION + 1)
    assert not hasattr(load_index(base), "cache_path")
    monkeypatch.undo()
    monkeypatch.setenv("MIRGE_LIB_CACHE", "0")
    assert not hasattr(load_index(base), "cache_path") and getattr(load_index(base), "cache_target", None) is None
    # a library directory that cannot be written to
    monkeypatch.setenv("MIRGE_LIB_CACHE", "1")
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "xdg"))
    os.unlink(path)
    ro = tmp_path / "ro"
    ro.mkdir()
    write_fasta(str(ro / "human_rrna.fa"), lib)
    os.chmod(ro, 0o555)
    try:
        if os.access(ro, os.W_OK):  # root ignores the mode bits: nothing to see here
            return
        p2 = libcache.save(str(ro / "human_rrna"), lib, packed, libcache.source_stamp(str(ro / "human_rrna")))
        assert p2.startswith(str(tmp_path / "xdg")) and load_index(str(ro / "human_rrna")).cache_path == p2
    finally:
        os.chmod(ro, 0o755)


def test_every_python_file_compiles_and_imports():
    """Every .py of the package, bench.py, __graft_entry__.py and tools/ byte-compiles, and every package module imports without a
    GPU (several are imported lazily by the CLI -- fastpath, a2i, gff --: a syntax error there would show only on the GPU box)."""
    import glob
    import importlib
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    files = glob.glob(os.path.join(root, "mirge3.0_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(root, "tools", "*.py")) + \
        glob.glob(os.path.join(root, "profiles", "*.py")) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(files) > 15
    for f in files:
        with open(f, "rb") as fh:
            compile(fh.read(), f, "exec")  # (raises SyntaxError; nothing is written next to the sources)
    import mirge3_amd  # noqa: F401
    for f in sorted(glob.glob(os.path.join(root, "mirge3.0_amd", "*.py"))):
        name = os.path.basename(f)[:-3]
        if name not in ("__init__", "__main__"):
            importlib.import_module("mirge3_amd." + name)

