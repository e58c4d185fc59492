"""The oracle's restatements of the A-to-I code (mirge2_tRF_a2i.py:230-518) and of create_gff (summary.py:48-606)
against what the REFERENCE's own functions returned (tests/golden/case4_gff_a2i, made by make_golden.py)."""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import GOLDEN

GFF_A2I_CASES = ["case4_gff_a2i", "case6_gff_a2i"]  # (case 6: round 4, the same recipe on other libraries and reads)


@pytest.mark.parametrize("case_name", GFF_A2I_CASES)
def test_a2i_restatement_equals_the_reference_functions(case_name):
    d = json.load(open(os.path.join(GOLDEN, case_name, "a2i_direct.json")))
    assert len(d["groups"]) >= 30
    n_true = n_pos = 0
    for g in d["groups"]:
        o = oracle.a2i_group(g["target"], g["reads"], g["counts"], set(g["retained"]))
        assert o["frame"] == g["aligned"], (g["target"], o["frame"], g["aligned"])
        assert o["states"] == g["states"]
        a = g["a2i"]
        assert o["kept"] == a["kept"] and o["positions"] == a["positions"]
        assert {str(k): v for k, v in o["count"].items()} == a["count"]
        assert o["countSumTrue"] == a["countSumTrue"] and o["seqCountTrue"] == a["seqCountTrue"]
        assert o["canonicalSeqCount"] == a["canonicalSeqCount"]
        for k, v in a["ratio"].items():
            assert o["ratio"][int(k)] == v
        for k, v in a["pvalue"].items():
            assert o["pvalue"][int(k)] == pytest.approx(v, rel=1e-12, abs=1e-300)
        assert o["census"] == g["mismatch_census"]
        n_true += sum(g["states"]); n_pos += len(a["positions"])
    assert n_true > 50 and n_pos >= 5


def _case4_gff_tables(case_name="case4_gff_a2i"):
    """name -> canonical sequence / precursor name -> precursor sequence, read the way summarize() reads them
    (summary.py:801-837): the mature FASTA, the GFF3 annotation, and `bowtie-inspect` of the hairpin index -- whose
    output ends with a newline, so that the LAST precursor's sequence is overwritten with '' (:819-826)."""
    from helpers import GoldenCase, ORG
    case = GoldenCase(case_name)
    lib = case.libdir
    mat, nm = {}, None
    for ln in open(f"{lib}/{ORG}/fasta.Libs/{ORG}_mature_miRBase.fa"):
        ln = ln.strip()
        if ln.startswith(">"):
            nm = ln[1:]
        else:
            mat[nm] = ln
    pre = dict(zip(case.libs["hairpin"].names, case.libs["hairpin"].seqs.to_list()))
    pre[case.libs["hairpin"].names[-1]] = ""
    pre_of, cur = {}, None
    for ln in open(f"{lib}/{ORG}/annotation.Libs/{ORG}_miRBase.gff3"):
        f = ln.rstrip("\n").split("\t")
        if len(f) < 9:
            continue
        if f[2] == "miRNA_primary_transcript":
            cur = f[8].split(";")[-1].replace("Name=", "")
        else:
            pre_of.setdefault(f[8].split(";")[2].replace("Name=", ""), cur)
    return case, mat, pre, pre_of


@pytest.mark.parametrize("case_name", GFF_A2I_CASES)
def test_gff_restatement_equals_the_reference_file(case_name):
    case, mat, pre, pre_of = _case4_gff_tables(case_name)
    n = 0
    kinds = set()
    for ln in open(os.path.join(case.dir, "sample_miRge3.gff")):
        if ln.startswith("#"):
            continue
        f = ln.rstrip("\n").split("\t")
        attrs = dict(x.split("=", 1) for x in f[8].split("; "))
        rec = oracle.gff_record(mat[f[0]], attrs["Read"], pre[pre_of[f[0]]])
        assert rec == (f[2], int(f[3]), int(f[4]), attrs["Variant"], attrs["Cigar"]), (f[0], mat[f[0]], attrs["Read"])
        kinds.add(attrs["Variant"].split(":")[0].split(",")[0])
        n += 1
    assert n > 650 and {"iso_5p", "iso_3p", "iso_add3p", "iso_add5p", "iso_snv_seed", "iso_snv", "NA"} <= kinds
