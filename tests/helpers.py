"""Shared test helpers: golden-case loader and oracle drivers."""
import csv
import os

import numpy as np

import mirge3_amd  # noqa: F401
from mirge3_amd.seqio import FlatSeqs, load_library_dir, load_merges

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["case1_single", "case2_two_samples", "case3_spikein", "case5_three_samples_spikein", "case7_two_samples_cr0.4"]
ORG, DB = "human", "miRBase"
PASS_LIBKEY = ["mirna", "hairpin", "mature_trna", "pre_trna", "snorna", "rrna", "ncrna_others",
               "mrna", "mirna", "spike-in"]
PASS_COLS = ["exact miRNA", "hairpin miRNA", "mature tRNA", "primary tRNA", "snoRNA", "rRNA",
             "ncrna others", "mRNA", "isomiR miRNA", "spike-in"]


class GoldenCase:
    def __init__(self, name):
        self.name = name
        self.dir = os.path.join(GOLDEN, name)
        self.spike = "spikein" in name
        self.cr = name.split("_cr")[1] if "_cr" in name else "0.1"  # -ex / --crThreshold the reference was run with
        self.n_pass = 10 if self.spike else 9
        self.libdir = os.path.join(self.dir, "libs")
        self.libs = load_library_dir(self.libdir, ORG, DB, with_spike=self.spike)
        self.merges = load_merges(self.libdir, ORG, DB)
        with open(os.path.join(self.dir, "collapsed_input.csv")) as fh:
            rows = list(csv.reader(fh))
        self.samples = rows[0][1:]
        self.seqs = [r[0] for r in rows[1:]]
        self.counts = np.array([[int(x) for x in r[1:]] for r in rows[1:]], dtype=np.int64)
        self.reads = FlatSeqs.from_list(self.seqs)
        self.counters = {}
        with open(os.path.join(self.dir, "counters.csv")) as fh:
            for r in csv.DictReader(fh):
                self.counters[r["sample"]] = r
        self.sample_read_counts = {s: int(self.counters[s]["total_input"]) for s in self.samples}
        self.trimmed = {s: int(self.counters[s]["trimmed_all"]) for s in self.samples}
        self.trimmed_unique = {s: int(self.counters[s]["trimmed_unique"]) for s in self.samples}

    def text(self, fname):
        with open(os.path.join(self.dir, fname)) as fh:
            return fh.read()

    def expected_annotation(self):
        """{sequence: (pass index, reference name)} from the reference's mapped.csv; sequences
        of unmapped.csv map to (-1, '')."""
        exp = {}
        with open(os.path.join(self.dir, "mapped.csv")) as fh:
            rd = csv.DictReader(fh)
            for r in rd:
                hits = [(p, r[c]) for p, c in enumerate(PASS_COLS) if c in r and r[c] != ""]
                assert len(hits) == 1, r
                assert r["annotFlag"] == "1"
                exp[r["Sequence"]] = hits[0]
        with open(os.path.join(self.dir, "unmapped.csv")) as fh:
            for r in csv.DictReader(fh):
                exp[r["Sequence"]] = (-1, "")
        return exp

    def oracle_libs(self):
        out = []
        for p in range(self.n_pass):
            lib = self.libs[PASS_LIBKEY[p]]
            out.append((lib.seqs.data, lib.seqs.offsets))
        return out

    def lib_of_pass(self, p):
        return self.libs[PASS_LIBKEY[p]]


def oracle_libs_from(libs, n_pass=9):
    return [(libs[PASS_LIBKEY[p]].seqs.data, libs[PASS_LIBKEY[p]].seqs.offsets) for p in range(n_pass)]


LEX_DIGIT = {"A": 1, "C": 2, "G": 3, "N": 4, "T": 5}


def lex_key0(seq: str) -> int:
    """the first word of the device's sort key of a read (csrc/kernels_csv.hpp, k_lexkey): 21 bases, 3 bits each (end 0, A 1, C 2,
    G 3, N 4, T 5), most significant first -- comparing these numbers is comparing the first 21 letters as Python strings"""
    key = 0
    for b in range(21):
        key = (key << 3) | (LEX_DIGIT[seq[b]] if b < len(seq) else 0)
    return key
