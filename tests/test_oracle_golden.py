"""The oracle against the golden vectors the REFERENCE produced (tests/golden/make_golden.py).

Pins: pass order, subset rules, T-tail strip, overwrite rule (manifoldAlign.py:12-146) and the
count join (summary.py) -- the reference's own code ran over the oracle's matcher to make these
files.  Does not pin bowtie's predicate itself ("parity unpinned", see oracle/mirge_oracle.c).
"""
import os

import numpy as np
import pytest

import oracle
from helpers import CASES, GoldenCase, GOLDEN


@pytest.fixture(scope="module", params=CASES)
def case(request):
    return GoldenCase(request.param)


def test_cascade_matches_reference_harness(case):
    for indexed in (False, True):
        ps, ref, off, mm = oracle.cascade(case.reads.data, case.reads.offsets, case.oracle_libs(),
                                          n_pass=case.n_pass, indexed=indexed)
        exp = case.expected_annotation()
        assert len(exp) == len(case.seqs)
        for i, s in enumerate(case.seqs):
            p = int(ps[i])
            name = case.lib_of_pass(p).names[int(ref[i])] if p >= 0 else ""
            assert (p, name) == exp[s], (s, indexed)


def test_join_matches_reference_summarize(case):
    ps, ref, off, mm = oracle.cascade(case.reads.data, case.reads.offsets, case.oracle_libs(),
                                      n_pass=case.n_pass)
    mir = case.libs["mirna"]
    j = oracle.join(ps, ref, case.counts, mir.names, mir.headers, case.merges, case.samples,
                    case.sample_read_counts, case.trimmed, case.trimmed_unique, cr_threshold=float(case.cr), spike=case.spike)
    assert j["report_csv"] == case.text("annotation.report.csv")
    assert j["counts_csv"] == case.text("miR.Counts.csv")
    assert j["rpm_csv"] == case.text("miR.RPM.csv")


def test_collapse_rule():
    seqs = ["ACGT" * 5, "TTTTACGTACGTACGTAC", "ACGT" * 5, "ACGTACGTACGTACGTACGN", "ACGT" * 5,
            "TTTTACGTACGTACGTAC", "ACGTACGTACGTACGTACG"]
    from mirge3_amd.seqio import FlatSeqs
    fs = FlatSeqs.from_list(seqs)
    first, cnt, inv = oracle.collapse(fs.data, fs.offsets)
    assert first.tolist() == [0, 1, 3, 6]
    assert cnt.tolist() == [3, 2, 1, 1]
    assert inv.tolist() == [0, 1, 0, 2, 0, 1, 3]
    f0, c0, i0 = oracle.collapse(fs.data[:0], fs.offsets[:1])
    assert len(f0) == 0 and len(i0) == 0


def test_umi_restatement_against_reference_vectors():
    """tests/golden/umi: slicing by the reference's own UMIParser, dict stages replayed around it."""
    import json
    with open(os.path.join(GOLDEN, "umi", "umi_cases.json")) as fh:
        cases = json.load(fh)
    raw = open(os.path.join(GOLDEN, "umi", "reads.txt")).read().split("\n")[:-1]
    assert len(cases) == 5
    for c in cases:
        f, b = c["front"], c["back"]
        for s, pure, tag in c["parser"]:
            assert oracle.umi_parser(s, f, b) == (pure, tag)
        for key, dedup in (("umi", False), ("udd", True)):
            d, trimmed, rows = oracle.umi_collapse(raw, f, b, c["min_len"], dedup)
            assert [list(x) for x in d] == c[key]["dict"]
            assert trimmed == c[key]["trimmed"] and len(d) == c[key]["unique"]
            assert ("".join(rows) if rows else None) == c[key]["umiCounts_csv"]


# the ten reads the reference's documentation prints for its two UMI command lines (docs/source/quick_start.md:290-298,
# 306-314): hsa-let-7a-5p with five distinct UMIs each
QIAGEN_READS = [
    "TGAGGTAGTAGGTTGTATAGTTAACTGTAGGCACCATCAATGTTAGACCTGCAAGATCGGAAGAGCACACGTCTG",
    "TGAGGTAGTAGGTTGTATAGTTAACTGTAGGCACCATCAATCAATGACGATTTAGATCGGAAGAGCACACGTCTG",
    "TGAGGTAGTAGGTTGTATAGTTAACTGTAGGCACCATCAATAAACAAAGATCCAGATCGGAAGAGCACACGTCTG",
    "TGAGGTAGTAGGTTGTATAGTTAACTGTAGGCACCATCAATCGCATCGCCGACAGATCGGAAGAGCACACGTCTG",
    "TGAGGTAGTAGGTTGTATAGTTAACTGTAGGCACCATCAATTTTGCCATTACTAGATCGGAAGAGCACACGTCTG",
]
ILLUMINA_4N_READS = [
    "TACATGAGGTAGTAGGTTGTATAGTTCCTCTGGAATTCTCGGGTGCCAAGGAACTCCAGTCACCGGAATATCTCG",
    "TACCTGAGGTAGTAGGTTGTATAGTTACTATGGAATTCTCGGGTGCCAAGGAACTCCAGTCACCGGAATATCTCG",
    "CAGGTGAGGTAGTAGGTTGTATAGTTGGTATGGAATTCTCGGGTGCCAAGGAACTCCAGTCACCGGAATATCTCG",
    "AGAATGAGGTAGTAGGTTGTATAGTTACTATGGAATTCTCGGGTGACAAGGAACTCCAGTCACCGGAATATCTCG",
    "AGGTTGAGGTAGTAGGTTGTATAGTTACTATGGAATTCTCGGGTGCCAAGGAACTCCAGTCACCGGAATATCTCG",
]
LET7A = "TGAGGTAGTAGGTTGTATAGTT"


def test_documented_umi_reads_are_known_answers():
    """The reference's two documented UMI command lines (quick_start.md:286,306) on the reads printed below them: the
    restated worker (digest.py:334-365) + baking's UMI stage (:164-205) must give let-7a-5p with five molecules."""
    recs = [(s, "I" * len(s)) for s in QIAGEN_READS for _ in range(3)]  # three PCR copies of every molecule
    opts = dict(q_back=10, adapter="AACTGTAGGCACCATCAAT")
    keys = oracle.umi_worker_reads(recs, opts, 0, 12, 16, qiagen=True)
    assert keys[0] == LET7A + "GTTAGACCTGCA" and len(keys) == 15
    d, trimmed, rows = oracle.umi_baking(keys, 0, 12, 16, dedup=True)
    assert d == [(LET7A, 5)] and trimmed == 5
    assert rows[1] == f"GTTAGACCTGCA,{LET7A},3\n" and len(rows) == 6
    d, trimmed, _ = oracle.umi_baking(keys, 0, 12, 16, dedup=False)
    assert d == [(LET7A, 15)] and trimmed == 15
    # -a illumina -umi 4,4 -udd; the fourth read's adapter carries a substitution and is still found
    recs = [(s, "I" * len(s)) for s in ILLUMINA_4N_READS for _ in range(2)]
    opts = dict(q_back=10, adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAG")
    keys = oracle.umi_worker_reads(recs, opts, 4, 4, 16, qiagen=False, per_modifier=False)
    assert keys[0] == "TACA" + LET7A + "CCTC"
    d, trimmed, rows = oracle.umi_baking(keys, 4, 4, 16, dedup=True)
    assert d == [(LET7A, 5)] and trimmed == 5 and rows[1] == f"TACACCTC,{LET7A},2\n"
    # at HEAD the worker counts inside its loop over the modifiers (digest.py:354-365): the quality-trimmed read, adapter
    # still on it, is a dictionary key of its own, and its 'insert' a row of the final table
    keys = oracle.umi_worker_reads(recs, opts, 4, 4, 16, qiagen=False, per_modifier=True)
    d, trimmed, _ = oracle.umi_baking(keys, 4, 4, 16, dedup=True)
    assert d[0] == (ILLUMINA_4N_READS[0][4:-4], 1) and d[1] == (LET7A, 5) and len(d) == 5 and trimmed == 10  # reads 2 and 5 share a 67-nt 'insert'


def test_qiagen_key_follows_the_references_string_rule():
    """digest.py:340-348 is string work on the untrimmed read: the FIRST occurrence of the trimmed read decides, the text
    up to its next occurrence is what the UMI is cut from, an empty trimmed read gives no UMI, a short remainder a short
    UMI -- and UMIParser then takes the missing bases from the insert's end."""
    ad, b = "AACTGTAGGCACCATCAAT", 12
    ins = "TGAGGTAGTAGGTTGTATAGTT"
    umi = "GTTAGACCTGCA"
    assert oracle.qiagen_key(ins + ad + umi + "AGATCGG", ins, len(ad), b) == ins + umi
    assert oracle.qiagen_key(ins, ins, len(ad), b) == ins                       # nothing trimmed: split -> ['', '']
    assert oracle.qiagen_key(ins + ad + umi[:5], ins, len(ad), b) == ins + (ad + umi[:5])[-12:]  # read ends inside the UMI
    assert oracle.qiagen_key("ACGT", "", len(ad), b) == ""                      # ValueError: empty separator
    rep = "ACGTACGTACGTACGTAC"                                                  # the trimmed read occurs twice:
    cur = rep + "GG" + rep + ad + umi                                           # [1] is what stands between the two
    assert oracle.qiagen_key(cur, rep, len(ad), b) == rep + "GG"
    # 5' quality trimming: the first occurrence may sit in front of the real one
    assert oracle.qiagen_key("TT" + rep + "CCC" + rep + ad, rep, len(ad), b) == rep + "CCC"
    # b == 0: Python's s[-0:] is the whole string -- the key keeps adapter and all
    assert oracle.qiagen_key(ins + ad + umi, ins, len(ad), 0) == ins + ad
    assert oracle.umi_parser(ins + umi[:5], 0, 12) == (ins[:-7], ins[-7:] + umi[:5])


# ---- the only bowtie outputs the reference itself holds (as comments next to the code that parses them)
def _fake_bowtie(args, cwd):
    import subprocess
    import sys
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fake_bowtie", "bowtie")
    r = subprocess.run([sys.executable, exe] + args, capture_output=True, text=True, cwd=cwd, timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_sam_line_quoted_by_the_reference(tmp_path):
    """mirge/libs/summary.py:1194 quotes a line of bowtie's SAM output for the mature-tRNA pass (`-v 1 -a --best --strata`):
    AAAACATCAGATTGTGAGTC at POS 18 of trnaMT_HisGTG_MT_+_12138_12206, MAPQ 255, 20M, XA:i:1 MD:Z:17A2 NM:i:1 -- one
    mismatch, reference base A under read position 17.  The stand-in behind the golden fixtures prints those fields
    for a reference built to hold that window at offset 17."""
    read = "AAAACATCAGATTGTGAGTC"
    window = "AAAACATCAGATTGTGAATC"
    ref = "GCGTTCCGTAGTCTAGC" + window + "CGGTACCATTGGA"
    (tmp_path / "trna.fa").write_text(f">trnaMT_HisGTG_MT_+_12138_12206\n{ref}\n>other\nGGGGCCCCAAAATTTTGGGGCCCCAAAATTTT\n")
    (tmp_path / "in.fa").write_text(f">{read}\n{read}\n")
    out = _fake_bowtie([str(tmp_path / "trna"), "-v", "1", "-f", "-a", "--best", "--strata", "--norc", "-S", "--threads", "1",
                        str(tmp_path / "in.fa")], tmp_path)
    line = [ln for ln in out.splitlines() if not ln.startswith("@")][0].split("\t")
    quoted = ("AAAACATCAGATTGTGAGTC 0 trnaMT_HisGTG_MT_+_12138_12206 18 255 20M * 0 0 AAAACATCAGATTGTGAGTC "
              "IIIIIIIIIIIIIIIIIIII XA:i:1 MD:Z:17A2 NM:i:1").split(" ")
    assert line == quoted  # (the quoted line ends with XM:i:2, whose meaning for an aligned read the manual does not settle)
    assert ref.index(window) == 17


def test_default_format_lines_quoted_by_the_reference(tmp_path):
    """mirge/libs/mirge2_tRF_a2i.py:1082-1085 quotes four lines of bowtie's default output for `-n 1 -f -a -3 2` against the
    genome: the read trimmed by two bases, '+' and '-' strand hits (the '-' ones print the reverse complement), 0-based
    offsets, and mismatch descriptors `14:T>G` = read offset from its 5' end : reference base > read base, both as they
    stand on the forward strand.  A four-chromosome genome built to hold those alignments gives those lines."""
    read = "AAAAACTGAGACTACTTTTG"
    fwd = "AAAAACTGAGACTACTTT"
    rc = "AAAGTAGTCTCAGTTTTT"
    pad = "GCGCGTACGCGC"
    genome = {"chr10": pad * 3 + fwd + pad, "chr2": pad + rc + pad * 2,
              "chr14": pad * 2 + rc[:3] + "T" + rc[4:] + pad, "chr4": pad * 4 + rc[:3] + "A" + rc[4:] + pad}
    (tmp_path / "g.fa").write_text("".join(f">{k}\n{v}\n" for k, v in genome.items()))
    (tmp_path / "in.fa").write_text(f">{read}\n{read}\n")
    out = _fake_bowtie([str(tmp_path / "g"), "-n", "1", "-f", "-a", "-3", "2", str(tmp_path / "in.fa")], tmp_path)
    got = sorted(ln.split("\t") for ln in out.splitlines())
    want = sorted([
        [read, "+", "chr10", str(len(pad) * 3), fwd, "I" * 18, "0", ""],
        [read, "-", "chr2", str(len(pad)), rc, "I" * 18, "0", ""],
        [read, "-", "chr14", str(len(pad) * 2), rc, "I" * 18, "0", "14:T>G"],
        [read, "-", "chr4", str(len(pad) * 4), rc, "I" * 18, "0", "14:A>G"],
    ])
    assert got == want
    # what the reference does with them (mirge2_tRF_a2i.py:1076-1096): the read has two 0-mismatch hits -> not retained
    counts = [ln[-1].count(":") for ln in got]
    assert sorted(counts) == [0, 0, 1, 1]
