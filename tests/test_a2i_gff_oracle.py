"""The oracle's restatements of the A-to-I code (mirge2_tRF_a2i.py:230-518) and of create_gff (summary.py:48-606)
against what the REFERENCE's own functions returned (tests/golden/case4_gff_a2i, made by make_golden.py)."""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import GOLDEN

CASE4 = os.path.join(GOLDEN, "case4_gff_a2i")


def test_a2i_restatement_equals_the_reference_functions():
    d = json.load(open(os.path.join(CASE4, "a2i_direct.json")))
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
