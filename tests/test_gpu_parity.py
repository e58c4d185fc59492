"""GPU parity tests: the HIP path, called through the C ABI (mirge3_amd._ffi), against the oracle
on the same seeded inputs and against the golden vectors the reference produced.

Bar: bit-exact.  Everything on this path is integer / index work; the only floating point is
RPM (rounded to 4 decimals, summary.py:768) and it is compared as text against the reference's CSV.
"""
import csv
import os
from collections import Counter
from types import SimpleNamespace

import numpy as np
import pytest

import oracle
from helpers import CASES, GoldenCase, ORG, DB, PASS_COLS, oracle_libs_from
import mirge3_amd  # noqa: F401
from mirge3_amd import _ffi, synth
from mirge3_amd.cascade import Cascade, PASSES, bwt_align
from mirge3_amd.collapse import baking, collapse_samples
from mirge3_amd.countjoin import summarize, summarize_device
from mirge3_amd.seqio import FlatSeqs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _ffi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ci_libs():
    return synth.make_libraries(seed=77, scale="ci")


@pytest.fixture(scope="module")
def ci_cascade(ctx, ci_libs):
    c = Cascade(ctx, ci_libs.libs)
    yield c
    c.close()


def _assert_same(o, g):
    for a, b, nm in zip(o, g, ("pass", "ref", "off", "mm")):
        bad = np.nonzero(a.astype(np.int64) != b.astype(np.int64))[0]
        assert bad.size == 0, (nm, bad[:5], a[bad[:5]], b[bad[:5]])


# ---------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("name", CASES)
def test_golden_cascade_and_join(ctx, name, tmp_path):
    case = GoldenCase(name)
    casc = Cascade(ctx, case.libs, spike_in=case.spike)
    dr = _ffi.DeviceReads.pack(ctx, case.reads)
    dr.set_counts(case.counts.astype(np.uint32))
    res = casc.run(dr)
    ps, ref, off, mm = res.fetch()
    exp = case.expected_annotation()
    for i, s in enumerate(case.seqs):
        p = int(ps[i])
        nm = case.lib_of_pass(p).names[int(ref[i])] if p >= 0 else ""
        assert (p, nm) == exp[s], s
    summarize_device(ctx, dr, res, case.libs["mirna"], case.merges, case.samples, case.sample_read_counts,
                     case.trimmed, case.trimmed_unique, float(case.cr), case.spike, workDir=tmp_path)
    for f in ("annotation.report.csv", "miR.Counts.csv", "miR.RPM.csv"):
        assert (tmp_path / f).read_text() == case.text(f), f
    # the packed reads survive a round trip
    assert dr.unpack().to_list() == case.seqs
    res.close(); dr.close(); casc.close()


@pytest.mark.parametrize("name", CASES)
def test_golden_dropin_signatures(name, tmp_path):
    """bwt_align / summarize with the reference's signatures on the golden library directory:
    mapped.csv, unmapped.csv and the three tables must equal what the reference wrote."""
    import pandas as pd
    case = GoldenCase(name)
    args = SimpleNamespace(threads=1, bowtie_path=None, bowtieVersion="True", quiet=True, bam_out=False,
                           tRNA_frag=False, spikeIn=case.spike, organism_name=ORG, libraries_path=case.libdir,
                           crThreshold=case.cr, gff_out=False, isoform_entropy=True, AtoI=False)
    df = pd.DataFrame(case.counts, columns=case.samples, index=pd.Index(case.seqs, name="Sequence"))
    df = df.assign(**dict.fromkeys(PASS_COLS, ''))
    df = df.assign(annotFlag=0).reindex(columns=['annotFlag'] + PASS_COLS + case.samples)
    out = bwt_align(args, df, str(tmp_path), DB)
    mapped, unmapped = out[out.annotFlag.eq(1)], out[out.annotFlag.eq(0)]
    mapped.to_csv(tmp_path / "mapped.csv")
    unmapped.to_csv(tmp_path / "unmapped.csv")
    assert (tmp_path / "mapped.csv").read_text() == case.text("mapped.csv")
    assert (tmp_path / "unmapped.csv").read_text() == case.text("unmapped.csv")
    summarize(args, str(tmp_path), DB, case.samples, mapped, case.sample_read_counts, case.trimmed,
              case.trimmed_unique)
    for f in ("annotation.report.csv", "miR.Counts.csv", "miR.RPM.csv", "isomirs.csv", "isomirs.samples.csv"):
        assert (tmp_path / f).read_text() == case.text(f), f


def test_bowtie_shim_process_boundary(tmp_path):
    """mirge3.0_amd/shim/{bowtie,bowtie-inspect}: the reference's process boundary on the MI355X engine
    (SURVEY.md 8b item 3).  The harness below replays what manifoldAlign.py:12-64,84-135 does around that
    boundary -- argv strings verbatim, FASTA named by sequence, SAM consumed field 0 / field 2 -- and the
    result must be the mapped.csv / unmapped.csv the reference itself wrote for the golden case."""
    import re
    import subprocess
    import sys
    import pandas as pd
    case = GoldenCase("case3_spikein")
    shim = os.path.join(os.path.dirname(os.path.abspath(mirge3_amd.__file__)), "shim")
    v = subprocess.run([sys.executable, os.path.join(shim, "bowtie"), "--version"], capture_output=True, text=True, check=True)
    assert v.stdout.split("\n")[0].split(" ")[2] in ["1.0.0", "1.2.1", "1.2.2", "1.2.3", "1.3.0", "1.3.1", "1.3.2"]
    idxdir = os.path.join(case.libdir, ORG, "index.Libs")
    n = subprocess.run([sys.executable, os.path.join(shim, "bowtie-inspect"), "-n", os.path.join(idxdir, f"{ORG}_mirna_{DB}")],
                       capture_output=True, text=True, check=True)
    assert n.stdout.strip().split("\n") == case.libs["mirna"].headers
    fa = subprocess.run([sys.executable, os.path.join(shim, "bowtie-inspect"), os.path.join(idxdir, f"{ORG}_hairpin_{DB}")],
                        capture_output=True, text=True, check=True).stdout.split("\n")
    assert max(len(x) for x in fa) <= max(60, max(len(h) + 1 for h in case.libs["hairpin"].headers))
    assert "".join(x for x in fa if not x.startswith(">")) == "".join(case.libs["hairpin"].seqs.to_list())

    df = pd.DataFrame(case.counts, columns=case.samples, index=pd.Index(case.seqs, name="Sequence"))
    df = df.assign(**dict.fromkeys(PASS_COLS, '')).assign(annotFlag=0).reindex(columns=['annotFlag'] + PASS_COLS + case.samples)
    suffix = ['_mirna_' + DB, '_hairpin_' + DB, '_mature_trna', '_pre_trna', '_snorna', '_rrna', '_ncrna_others', '_mrna',
              '_mirna_' + DB, '_spike-in']
    fasta = tmp_path / "bwtInput.fasta"
    for it in range(10):
        if it == 0:
            recs = [(q, q) for q in df.index if len(q) < 26]
        elif it == 1:
            recs = [(q, q) for q in df.index if len(q) > 25]
        else:
            un = list(df.index[df.annotFlag.eq(0)])
            if it == 3:
                recs = [(q, q[:re.search('T{3,}$', q).start()]) for q in un if re.search('T{3,}$', q)]
            else:
                recs = [(q, q) for q in un]
        fasta.write_text("".join(f">{q}\n{x}\n" for q, x in recs))
        cmd = os.path.join(shim, "bowtie") + " " + os.path.join(idxdir, ORG + suffix[it]) + PASSES[it][2] + "2 " + str(fasta)
        run = subprocess.run(cmd, shell=True, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        body = [ln.split("\t") for ln in run.stdout.split("\n") if ln and not ln.startswith("@")]
        assert [b[0] for b in body] == [q for q, _ in recs]  # one line per read, input order
        for f in body:
            assert len(f) >= 11
            if f[2] != "*":
                df.at[f[0], PASS_COLS[it]] = f[2]
                df.at[f[0], 'annotFlag'] = 1
                lib, r = case.lib_of_pass(it), case.lib_of_pass(it).names.index(f[2])
                window = lib.seqs.get(r)[int(f[3]) - 1:int(f[3]) - 1 + len(f[9])]
                nm = int([t for t in f[11:] if t.startswith("NM:i:")][0][5:])
                assert len(window) == len(f[9]) and sum(a != b for a, b in zip(window, f[9])) == nm and f[5] == f"{len(f[9])}M"
    df = df.fillna('')
    mapped, unmapped = df[df.annotFlag.eq(1)], df[df.annotFlag.eq(0)]
    mapped.to_csv(tmp_path / "mapped.csv")
    unmapped.to_csv(tmp_path / "unmapped.csv")
    assert (tmp_path / "mapped.csv").read_text() == case.text("mapped.csv")
    assert (tmp_path / "unmapped.csv").read_text() == case.text("unmapped.csv")


def _crosscheck(bowtie_dir, n=4000):
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bowtie_crosscheck.py"), "--bowtie-dir", bowtie_dir, "--synthetic", str(n)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "membership identical in every pass" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def test_bowtie_crosscheck_harness(tmp_path):
    """The harness of tools/bowtie_crosscheck.py (argv strings, FASTA naming, SAM parsing, per-pass comparison) run
    against the shim.  This checks the HARNESS, not the predicate: the shim answers from the GPU engine."""
    _crosscheck(os.path.join(os.path.dirname(os.path.abspath(mirge3_amd.__file__)), "shim"))


def test_bowtie_crosscheck_real_bowtie(tmp_path):
    """The alignment predicate against a REAL bowtie 1.x: skipped, not passed, where none is installed -- the
    predicate then stays 'parity unpinned' (DESIGN.md section 3).  With a bowtie on PATH or in MIRGE_BOWTIE_DIR the
    per-pass membership must be identical and the tool's report is kept under gpurun_out/."""
    import shutil
    real = os.environ.get("MIRGE_BOWTIE_DIR") or (os.path.dirname(shutil.which("bowtie-build")) if shutil.which("bowtie-build") else None)
    if not real:
        pytest.skip("no bowtie 1.x (bowtie + bowtie-build) on this box: the alignment predicate stays unpinned")
    out = _crosscheck(real, 20000)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "bowtie_crosscheck_real.txt"), "w") as fh:
        fh.write(out)


def test_cutadapt_crosscheck_real_cutadapt():
    """tools/cutadapt_crosscheck.py: the trimming restatement (oracle and k_trim) against a REAL cutadapt on ten option
    sets -- skipped, not passed, where none is installed (row N4 then stays 'parity unpinned')."""
    import shutil
    import subprocess
    import sys
    if shutil.which("cutadapt") is None:
        pytest.skip("no cutadapt on this box: the trimming restatement stays unpinned")
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "cutadapt_crosscheck.py"), "--gpu"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:]


def test_shim_prints_the_sam_line_the_reference_quotes(tmp_path):
    """mirge/libs/summary.py:1194 quotes bowtie's SAM line for a mature-tRNA hit with one mismatch (the only SAM output
    the reference holds): POS 18, MAPQ 255, 20M, XA:i:1 MD:Z:17A2 NM:i:1.  The MI355X shim, asked the same way
    (`-v 1 -f -a --best --strata --norc -S`), prints those fields for a reference that holds the window at offset 17."""
    import subprocess
    import sys
    shim = os.path.join(os.path.dirname(os.path.abspath(mirge3_amd.__file__)), "shim", "bowtie")
    read, window = "AAAACATCAGATTGTGAGTC", "AAAACATCAGATTGTGAATC"
    ref = "GCGTTCCGTAGTCTAGC" + window + "CGGTACCATTGGA"
    (tmp_path / "trna.fa").write_text(f">trnaMT_HisGTG_MT_+_12138_12206\n{ref}\n>other\nGGGGCCCCAAAATTTTGGGGCCCCAAAATTTT\n")
    (tmp_path / "in.fa").write_text(f">{read}\n{read}\n")
    r = subprocess.run([sys.executable, shim, str(tmp_path / "trna"), "-v", "1", "-f", "-a", "--best", "--strata", "--norc", "-S",
                        "--threads", "1", str(tmp_path / "in.fa")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if not ln.startswith("@")][0].split("\t")
    assert line == ("AAAACATCAGATTGTGAGTC 0 trnaMT_HisGTG_MT_+_12138_12206 18 255 20M * 0 0 AAAACATCAGATTGTGAGTC "
                    "IIIIIIIIIIIIIIIIIIII XA:i:1 MD:Z:17A2 NM:i:1").split(" ")


# ---------------------------------------------------------------- oracle on seeded inputs
def test_cascade_vs_oracle_ci_scale(ctx, ci_libs, ci_cascade):
    reads = synth.make_reads(ci_libs, 60000, seed=5, n_frac=0.01)
    rng = np.random.default_rng(3)
    extra = []  # W=2 / W=4 width groups incl. the 128-nt maximum
    for key in ("mrna", "ncrna_others", "rrna"):
        lib = ci_libs.libs[key]
        for L in list(rng.integers(51, 129, size=80)) + [31, 32, 33, 64, 65, 128]:
            s = lib.seqs.get(int(rng.integers(0, len(lib))))
            if len(s) <= L:
                continue
            a = int(rng.integers(0, len(s) - L))
            x = list(s[a:a + int(L)])
            for _ in range(int(rng.integers(0, 3))):
                q = int(rng.integers(0, len(x)))
                x[q] = "ACGT"[("ACGT".index(x[q]) + 1) % 4] if x[q] in "ACGT" else "A"
            extra.append("".join(x))
    short = []  # below the default --minimum-length: 1..15 nt, with and without a T tail
    for key in ("mirna", "pre_trna", "mature_trna", "snorna", "mrna"):
        lib = ci_libs.libs[key]
        for L in range(1, 16):
            for _ in range(6):
                s = lib.seqs.get(int(rng.integers(0, len(lib))))
                a = int(rng.integers(0, len(s) - L))
                x = list(s[a:a + L])
                if rng.random() < 0.4 and x[0] in "ACGT":
                    q = int(rng.integers(0, L)); x[q] = "ACGT"[("ACGT".index(x[q]) + 1) % 4] if x[q] in "ACGT" else "A"
                short += ["".join(x), "".join(x) + "T" * int(rng.integers(3, 7))]
    short += ["A", "T", "TTT", "TTTT", "ATTT", "N", "NNNN", "ACGTTTT", "G" * 15, "AC" * 7]
    allr = FlatSeqs.from_list(reads.to_list() + extra + short + ["T" * 16, "T" * 40, "A" * 16, "ACGTN" * 4])
    g = ci_cascade.annotate(allr)
    o = oracle.cascade(allr.data, allr.offsets, oracle_libs_from(ci_libs.libs), n_pass=9, indexed=True)
    _assert_same(o, g)
    assert (g[0] >= 0).mean() > 0.5


def test_low_complexity_libraries(ctx):
    """Repeats make probe buckets thousands of entries long (poly-A tails, dinucleotide repeats): this
    is the wave-cooperative branch of align_hybrid, which random libraries hardly reach."""
    rng = np.random.default_rng(9)
    base = synth.make_libraries(seed=31, scale="tiny").libs
    libs = dict(base)
    for key, n in (("mrna", 60), ("ncrna_others", 40), ("snorna", 20), ("mirna", 0)):
        lib = libs[key]
        seqs = lib.seqs.to_list()
        for i in range(n):
            body = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(40, 200))))
            kind = i % 4
            rep = {0: "A" * int(rng.integers(30, 300)), 1: "CA" * int(rng.integers(20, 150)),
                   2: "T" * int(rng.integers(30, 120)), 3: "GGC" * int(rng.integers(10, 60))}[kind]
            seqs.append(body + rep + ("" if i % 2 else body[:20]))
        from mirge3_amd.seqio import Library
        libs[key] = Library(lib.names + [f"rep{key}{i}" for i in range(n)], FlatSeqs.from_list(seqs))
    reads = []
    for L in list(range(16, 51)) + [64, 100, 128]:
        for unit in ("A", "CA", "T", "GGC", "AC"):
            s = (unit * 200)[:L]
            reads.append(s)
            for _ in range(3):  # 1-2 substitutions
                x = list(s)
                for _ in range(int(rng.integers(1, 3))):
                    q = int(rng.integers(0, L)); x[q] = "ACGT"[("ACGT".index(x[q]) + int(rng.integers(1, 4))) % 4]
                reads.append("".join(x))
    reads += synth.make_reads(synth.make_libraries(seed=31, scale="tiny"), 2000, seed=4).to_list()
    fs = FlatSeqs.from_list(reads)
    casc = Cascade(ctx, libs)
    g = casc.annotate(fs)
    o = oracle.cascade(fs.data, fs.offsets, oracle_libs_from(libs), n_pass=9, indexed=False)
    _assert_same(o, g)
    assert (g[0] >= 0).sum() > 300
    casc.close()


def test_repeat_rich_libraries_vs_oracle(ctx):
    """Libraries with the repeat structure real ones have (synth.make_libraries(repeats=True): poly-A tails on 60 % of the mRNAs,
    Alu-like families at 5-15 % divergence, simple repeats, tRNA isodecoder families) and 40 k reads of synth.REPEAT_MIX (15 % poly-A /
    poly-T / simple-repeat / Alu-derived reads with 0-2 errors): every field of every read against the oracle -- its k-mer variant on all
    of them, brute force on 6 k -- through the one-call route and the staged one."""
    sl = synth.make_libraries(seed=123, scale="ci", repeats=True)
    assert sl.repeat_families and len(sl.repeat_families[0]) == 300
    mr = sl.libs["mrna"].seqs.to_list()
    assert sum(q.endswith("A" * 20) for q in mr) > 0.4 * len(mr)
    reads = synth.make_reads(sl, 40000, seed=11, mix=synth.REPEAT_MIX, n_frac=0.005)
    casc = Cascade(ctx, sl.libs)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq, res = casc.collapse_and_run(raw)
    g = res.fetch()
    useq = uniq.unpack()
    o = oracle.cascade(useq.data, useq.offsets, oracle_libs_from(sl.libs), n_pass=9, indexed=True)
    _assert_same(o, g)
    sub = useq.take(np.arange(0, len(useq), max(1, len(useq) // 6000)))
    _assert_same(oracle.cascade(sub.data, sub.offsets, oracle_libs_from(sl.libs), n_pass=9, indexed=False), casc.annotate(sub))
    # the repeat-derived reads do land in the libraries that carry the repeats
    ps = g[0]
    assert (ps == 7).sum() > 0.03 * len(useq) and (ps >= 0).sum() > 0.5 * len(useq)
    res.close(); uniq.close(); raw.close(); casc.close()


def test_references_shorter_than_a_resolve_granule(ctx):
    """Position -> (reference, offset) reads one entry per 16 positions (ResolveTable); references of 0-15 nt between
    the real ones put several starts into a granule -- the entry's search branch -- and a start on every offset of one."""
    from mirge3_amd.seqio import Library
    rng = np.random.default_rng(77)
    sl = synth.make_libraries(seed=33, scale="tiny")
    libs = dict(sl.libs)
    for key in ("mirna", "hairpin", "mature_trna", "mrna"):
        lib = libs[key]
        names, seqs = [], []
        for i, (nm, sq) in enumerate(zip(lib.names, lib.seqs.to_list())):
            for k in range(int(rng.integers(0, 4))):
                names.append(f"tiny{key}{i}_{k}")
                seqs.append("".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(0, 16)))))
            names.append(nm)
            seqs.append(sq)
        libs[key] = Library(names, FlatSeqs.from_list(seqs))
    reads = synth.make_reads(sl, 6000, seed=5, n_frac=0.02)
    casc = Cascade(ctx, libs)
    g = casc.annotate(reads)
    o = oracle.cascade(reads.data, reads.offsets, oracle_libs_from(libs), n_pass=9, indexed=False)
    _assert_same(o, g)
    assert (g[0] >= 0).mean() > 0.5
    casc.close()


def test_cascade_prepare_builds_the_tables_ahead(ctx, ci_libs):
    """mirge_cascade_prepare: the first cascade's table construction as a call of its own (the CLI times it as a stage);
    same annotation with and without it, from raw reads (only their lengths matter) and from reads without a histogram."""
    reads = synth.make_reads(ci_libs, 4000, seed=23, n_frac=0.02)
    want = None
    for ahead in (False, True):
        casc = Cascade(ctx, ci_libs.libs)
        raw = _ffi.DeviceReads.pack(ctx, reads)
        if ahead:
            casc.prepare(raw)
            casc.prepare(raw)  # a no-op the second time
        uniq, res = casc.collapse_and_run(raw)
        got = (uniq.unpack().to_list(), [a.tolist() for a in res.fetch()])
        order = np.argsort(np.array(got[0], dtype=object), kind="stable")
        got = ([got[0][i] for i in order], [[a[i] for i in order] for a in got[1]])
        if want is None:
            want = got
        else:
            assert got == want
        res.close(); uniq.close(); raw.close(); casc.close()


def test_cascade_vs_bruteforce_oracle(ctx, ci_libs, ci_cascade):
    reads = synth.make_reads(ci_libs, 3000, seed=17, n_frac=0.03)
    g = ci_cascade.annotate(reads)
    o = oracle.cascade(reads.data, reads.offsets, oracle_libs_from(ci_libs.libs), n_pass=9, indexed=False)
    _assert_same(o, g)


def test_collapse_vs_oracle(ctx, ci_libs):
    reads = synth.make_reads(ci_libs, 200000, seed=8, n_frac=0.01, pool=20000)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq = raw.collapse()
    counts, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    o_first, o_cnt, o_inv = oracle.collapse(reads.data, reads.offsets)
    assert len(uniq) == len(o_first)
    order = np.argsort(first, kind="stable")
    assert np.array_equal(first[order], o_first)
    assert np.array_equal(counts[order, 0].astype(np.int64), o_cnt)
    allr = reads.to_list()
    assert [seqs[i] for i in order] == [allr[i] for i in o_first]
    assert int(counts.sum()) == len(reads)
    raw.close(); uniq.close()


def test_collapse_multi_sample(ctx, ci_libs):
    samples = [synth.make_reads(ci_libs, 30000, seed=40 + s, pool=4000) for s in range(3)]
    uniq = collapse_samples(ctx, samples)
    counts, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    exp = [Counter(s.to_list()) for s in samples]
    union = set().union(*[set(c) for c in exp])
    assert set(seqs) == union and len(seqs) == len(union)
    for i, s in enumerate(seqs):
        assert [int(x) for x in counts[i]] == [c.get(s, 0) for c in exp]
    uniq.close()


def test_edge_cases(ctx, ci_libs, ci_cascade):
    # empty set
    empty = FlatSeqs(np.zeros(0, np.uint8), np.zeros(1, np.int64))
    dr = _ffi.DeviceReads.pack(ctx, empty)
    u = dr.collapse()
    assert len(u) == 0
    res = ci_cascade.run(u)
    assert res.fetch()[0].shape == (0,)
    cls, ex, iso = _ffi.count_join(ctx, u, res, 0, 8, len(ci_libs.libs["mirna"]))
    assert cls.sum() == 0 and ex.sum() == 0
    res.close(); u.close(); dr.close()
    # nothing annotated
    rnd = FlatSeqs.from_list(["ACGTTGCAAGCTTGCAAGGC"[::-1] * 1, "GGGGGGGGGGGGGGGGGGGGGGGG"])
    ps = ci_cascade.annotate(rnd)[0]
    o = oracle.cascade(rnd.data, rnd.offsets, oracle_libs_from(ci_libs.libs), n_pass=9)
    assert np.array_equal(ps.astype(np.int32), o[0])
    # loud failures: only beyond what a 16-bit length holds (reads of 256-65535 nt are the long class: test_reads_longer_than_255_nt)
    with pytest.raises(RuntimeError, match="limit is 65535"):
        _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list(["A" * 65536]))
    with pytest.raises(RuntimeError, match="limit is 65535"):
        _ffi.DeviceReads.parse(ctx, b"@r\n" + b"A" * 70000 + b"\n+\n" + b"I" * 70000 + b"\n", 1, 16)
    # untrimmed Illumina lengths (150 nt) go through: the fourth width class
    long_ref = ci_libs.libs["mrna"].seqs.get(5)
    lr = [long_ref[40:190], long_ref[100:355], long_ref[7:136]]
    g = ci_cascade.annotate(FlatSeqs.from_list(lr))
    o = oracle.cascade(FlatSeqs.from_list(lr).data, FlatSeqs.from_list(lr).offsets, oracle_libs_from(ci_libs.libs), n_pass=9)
    assert all(np.array_equal(x.astype(np.int64), y.astype(np.int64)) for x, y in zip(g, o)) and (g[0] == 7).all()
    # malformed texts are refused, not shifted: a truncated FASTQ record, a blank line inside, a wrapped FASTA
    for bad in (b"@r1\nACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIII\n@r2\nACGT\n", b"@r1\nACGTACGTACGTACGTAC\n\n+\nIIIIIIIIIIIIIIIIII\n",
                b"@r1\nACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIII\nr2\nACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIII\n",
                b">a\nACGTACGTACGT\nACGTACGT\n>b\nACGTACGTACGTACGTAC\n"):
        with pytest.raises(RuntimeError, match="record"):
            _ffi.DeviceReads.parse(ctx, bad, 0, 16)
    # blank lines at the end of the file are not records
    r_ok, n_rec = _ffi.DeviceReads.parse(ctx, b"@r1\nACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIII\n\n\r\n", 0, 16)
    assert n_rec == 1 and len(r_ok) == 1
    r_ok.close()
    # a wrapped FASTA through the host-side unwrap
    from mirge3_amd.collapse import unwrap_fasta
    r_ok, n_rec = _ffi.DeviceReads.parse(ctx, unwrap_fasta(b">a\nACGTACGTACGT\nACGTACGT\n>b x\nACGTACGTACGTACGTAC\r\nGG\r\n"), 0, 16)
    assert n_rec == 2 and r_ok.unpack().to_list() == ["ACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTACGG"]
    r_ok.close()
    with pytest.raises(RuntimeError, match="no nucleotide code"):
        _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list(["ACGT5*ACGTACGTACGT"]))
    # IUPAC ambiguity codes are what bowtie makes of them: N (flagged on the read set)
    iu = _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list(["ACGTRYACGTACGTACGT", "ACGTACGTACGTACGTAC"]))
    assert iu.iupac_seen and iu.unpack().to_list() == ["ACGTNNACGTACGTACGT", "ACGTACGTACGTACGTAC"]
    iu.close()
    iu, _ = _ffi.DeviceReads.parse(ctx, b"ACGTACGTKCGTACGTAC\nACGTACGTACGTACGTAC\n", 3, 16)
    assert iu.iupac_seen and iu.unpack().to_list()[0] == "ACGTACGTNCGTACGTAC"
    iu.close()
    plain = _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list(["ACGTNACGTACGTACGTA"]))
    assert not plain.iupac_seen
    plain.close()
    # '.', the no-call of Illumina's old pipelines (bowtie reads it as N): the same, through both entry points and in a long read
    for dots in (_ffi.DeviceReads.pack(ctx, FlatSeqs.from_list(["ACGT.ACGTACGTACGT.", "A" * 299 + "."])),
                 _ffi.DeviceReads.parse(ctx, b"@a\nACGT.ACGTACGTACGT.\n+\nIIIIIIIIIIIIIIIIII\n@b\n" + b"A" * 299 + b".\n+\n" + b"I" * 300 + b"\n", 1, 16)[0]):
        assert dots.iupac_seen and dots.unpack().to_list() == ["ACGTNACGTACGTACGTN", "A" * 299 + "N"]
        dots.close()
    # lower-case and U are accepted as their upper-case / T
    a = ci_cascade.annotate(FlatSeqs.from_list([ci_libs.libs["mirna"].seqs.get(3).lower().replace("t", "u")]))
    b = ci_cascade.annotate(FlatSeqs.from_list([ci_libs.libs["mirna"].seqs.get(3)]))
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and a[0][0] == 0


def test_baking_dropin(ctx, ci_libs, tmp_path):
    """FASTQ files in, the reference's DataFrame schema + counters out (digest.py:237-261,212-217)."""
    names, files, exp = [], [], []
    for s in range(2):
        reads = synth.make_reads(ci_libs, 5000, seed=60 + s, pool=700).to_list() + ["ACGTACGT", "ACGTACGTACGTACG"]
        p = tmp_path / f"S{s + 1}.fastq"
        with open(p, "w") as fh:
            for i, r in enumerate(reads):
                fh.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
        names.append(f"S{s + 1}"); files.append(str(p)); exp.append(reads)
    args = SimpleNamespace(quiet=True, minimum_length=16, adapters=None, front=None, uniq_mol_ids=None)
    df, src, trimmed, uniq = baking(args, files, names, str(tmp_path), ctx=ctx)
    cnt = [Counter(r for r in e if len(r) >= 16) for e in exp]
    union = sorted(set(cnt[0]) | set(cnt[1]))
    assert list(df.index) == union and df.index.name == "Sequence"
    assert list(df.columns) == ["annotFlag"] + PASS_COLS + names
    for s, nm in enumerate(names):
        assert df[nm].tolist() == [cnt[s].get(q, 0) for q in union]
        assert src[nm] == len(exp[s]) and trimmed[nm] == sum(cnt[s].values()) and uniq[nm] == len(cnt[s])
    assert (df.annotFlag == 0).all() and (df[PASS_COLS] == "").all().all()
    # single sample keeps first-appearance order
    df1, *_ = baking(args, files[:1], names[:1], str(tmp_path), ctx=ctx)
    seen = list(dict.fromkeys(r for r in exp[0] if len(r) >= 16))
    assert list(df1.index) == seen
    with pytest.raises(NotImplementedError):  # up to two adapters per run (AdapterCutter's best match of two); three are refused
        baking(SimpleNamespace(quiet=True, minimum_length=16, adapters=[("back", "TGGAATTC"), ("back", "AGATCGG")], front=[("front", "ACGT")],
                               uniq_mol_ids=None), files, names, str(tmp_path), ctx=ctx)
    with pytest.raises(RuntimeError, match="adapter characters"):  # letters that are no bases reach the C ABI and fail there
        baking(SimpleNamespace(quiet=True, minimum_length=16, adapters=[("back", "ACGTZ")], front=None, uniq_mol_ids=None),
               files, names, str(tmp_path), ctx=ctx)
    with pytest.raises(NotImplementedError, match="non-internal"):  # cutadapt's `ADAPTERX`: a form of its own, refused by name
        baking(SimpleNamespace(quiet=True, minimum_length=16, adapters=[("back", "ACGTX")], front=None, uniq_mol_ids=None),
               files, names, str(tmp_path), ctx=ctx)


def test_device_text_parser_equals_host_parser(ctx, ci_libs, tmp_path):
    """mirge_reads_parse (records found, length filter, grouping and 2-bit packing on the GPU) against the host
    parser + mirge_reads_pack on the same text: FASTQ with CRLF and no final newline, single-line FASTA, one
    sequence per line, reads with N, reads of 32-128 nt, records shorter than --minimum-length, empty input."""
    from mirge3_amd.collapse import read_fastq_sequences, filter_min_length
    rng = np.random.default_rng(8)
    reads = synth.make_reads(ci_libs, 30000, seed=77, n_frac=0.02).to_list()
    longs = ["".join("ACGT"[c] for c in rng.integers(0, 4, size=int(L))) for L in list(rng.integers(32, 129, size=400)) + [31, 32, 64, 65, 128]]
    shorts = ["ACGT", "ACGTACGTACGTACG", "", "A"]
    seqs = reads + longs + shorts + [longs[3][:40] + "N" + longs[3][41:90], "acgtacgtacgtacgtnnu"]
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    texts = {
        "fastq": "".join(f"@r{i} x\n{q}\n+\n{'I' * len(q)}\n" for i, q in enumerate(seqs)),
        "fastq_crlf_noeol": "".join(f"@r{i}\r\n{q}\r\n+\r\n{'I' * len(q)}\r\n" for i, q in enumerate(seqs))[:-2],
        "fasta": "".join(f">r{i}\n{q}\n" for i, q in enumerate(seqs)),
        "lines": "\n".join(q for q in seqs if q) + "\n",
    }
    for name, text in texts.items():
        path = tmp_path / (name + ".txt")
        path.write_text(text, newline="")
        host = read_fastq_sequences(str(path))
        for min_len in (0, 16):
            exp = filter_min_length(host, min_len)
            dr, n_rec = _ffi.DeviceReads.parse(ctx, path.read_bytes(), 0, min_len)
            assert n_rec == len(host), (name, n_rec, len(host))
            assert len(dr) == len(exp), (name, min_len)
            assert dr.unpack().to_list() == [q.upper().replace("U", "T") for q in exp.to_list()], (name, min_len)
            hp = _ffi.DeviceReads.pack(ctx, exp)
            u1, u2 = dr.collapse(), hp.collapse()
            c1, f1 = u1.counts(); c2, f2 = u2.counts()
            o1, o2 = np.argsort(f1, kind="stable"), np.argsort(f2, kind="stable")
            assert np.array_equal(f1[o1], f2[o2]) and np.array_equal(c1[o1], c2[o2])
            s1, s2 = u1.unpack().to_list(), u2.unpack().to_list()
            assert [s1[i] for i in o1] == [s2[i] for i in o2]
            for h in (u1, u2, dr, hp):
                h.close()
    # mirge_reads_concat: the samples of a run appended on the device, handle order = part order
    a, _ = _ffi.DeviceReads.parse(ctx, texts["fastq"].encode(), 0, 16)
    b, _ = _ffi.DeviceReads.parse(ctx, texts["fasta"].encode()[:20000] + b"\n", 0, 0)
    e, _ = _ffi.DeviceReads.parse(ctx, b"", 0, 0)
    ab = _ffi.DeviceReads.concat(ctx, [a, e, b, a])
    assert ab.unpack().to_list() == a.unpack().to_list() + b.unpack().to_list() + a.unpack().to_list()
    sid = np.repeat(np.arange(4, dtype=np.int32), [len(a), 0, len(b), len(a)])
    u = ab.collapse(sid, 4)
    cnt, _ = u.counts()
    assert cnt[:, 0].sum() == len(a) and cnt[:, 1].sum() == 0 and cnt[:, 2].sum() == len(b) and np.array_equal(cnt[:, 0], cnt[:, 3])
    for h in (u, ab, a, b, e):
        h.close()
    dr, n_rec = _ffi.DeviceReads.parse(ctx, b"", 0, 16)
    assert len(dr) == 0 and n_rec == 0
    dr, n_rec = _ffi.DeviceReads.parse(ctx, b"@r\nACGT\n+\nIIII\n", 0, 16)
    assert len(dr) == 0 and n_rec == 1 and len(dr.collapse()) == 0
    dr, n_rec = _ffi.DeviceReads.parse(ctx, ("A" * 256 + "\n").encode(), 3, 0)  # one base beyond the widest templated class
    assert n_rec == 1 and dr.unpack().to_list() == ["A" * 256] and int(dr.group_counts()[4]) == 1
    dr.close()
    with pytest.raises(RuntimeError, match="no nucleotide code"):
        _ffi.DeviceReads.parse(ctx, b"ACGTACGTACGTACGTXACGT\n", 3, 0)
    # a record whose quality line is shorter / longer than its sequence line: dnaio raises, and so does the trimming parser
    # (k_trim would otherwise take the missing qualities from the next record's bytes)
    good = b"@a\nACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIII\n"
    for bad in (b"@b\nACGTACGTACGTACGTAC\n+\nIIIIIIIIIIII\n", b"@b\nACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIII\n"):
        for text in (good + bad, bad + good, good + bad + good):
            with pytest.raises(RuntimeError, match="quality line"):
                _ffi.DeviceReads.parse(ctx, text, 1, 16, _ffi.MirgeTrim.make(quality_back=10))
    dr, n_rec = _ffi.DeviceReads.parse(ctx, good.replace(b"\n", b"\r\n") * 3, 1, 16, _ffi.MirgeTrim.make(quality_back=10))
    assert len(dr) == 3 and n_rec == 3
    dr.close()


def test_baking_umi_against_reference_vectors(ctx, tmp_path):
    """-umi f,b [-udd] [-tcf] (digest.py:164-205,219-229,305-315): tests/golden/umi was written with the
    reference's own UMIParser; two samples check the joined matrix against the oracle's restatement."""
    import json
    from helpers import GOLDEN
    with open(os.path.join(GOLDEN, "umi", "umi_cases.json")) as fh:
        cases = json.load(fh)
    src_file = os.path.join(GOLDEN, "umi", "reads.txt")
    raw = open(src_file).read().split("\n")[:-1]
    for c in cases:
        for key, dedup in (("umi", False), ("udd", True)):
            wd = tmp_path / f"w{c['front']}_{c['back']}_{key}"
            wd.mkdir()
            args = SimpleNamespace(quiet=True, minimum_length=c["min_len"], adapters=None, front=None, tcf_out=True,
                                   uniq_mol_ids=f"{c['front']},{c['back']}", umiDedup=dedup)
            df, src, trimmed, uniq = baking(args, [src_file], ["reads"], str(wd), ctx=ctx)
            exp = c[key]
            assert [[q, n] for q, n in zip(df.index, df["reads"].tolist())] == exp["dict"]
            assert src["reads"] == c["total_input"] and trimmed["reads"] == exp["trimmed"] and uniq["reads"] == exp["unique"]
            if dedup:
                assert open(wd / "reads_umiCounts.csv").read() == exp["umiCounts_csv"]
            else:
                assert not (wd / "reads_umiCounts.csv").exists()
            d = dict(map(tuple, exp["dict"]))
            tcf = "".join(f">seq{k + 1}_{d[q]}\n{q}\n" for k, q in enumerate(sorted(d, key=d.get, reverse=True)))
            assert open(wd / "reads.trim.collapse.fa").read() == tcf
    # two samples, -udd: the union is sorted (digest.py:243), counts are molecules per sample
    half = tmp_path / "half.txt"
    half.write_text("\n".join(raw[::2]) + "\n")
    args = SimpleNamespace(quiet=True, minimum_length=16, adapters=None, front=None, uniq_mol_ids="4,4", umiDedup=True)
    df, src, trimmed, uniq = baking(args, [src_file, str(half)], ["A", "B"], str(tmp_path), ctx=ctx)
    da = dict(oracle.umi_collapse(raw, 4, 4, 16, True)[0])
    db = dict(oracle.umi_collapse(raw[::2], 4, 4, 16, True)[0])
    union = sorted(set(da) | set(db))
    assert list(df.index) == union
    assert df["A"].tolist() == [da.get(q, 0) for q in union] and df["B"].tolist() == [db.get(q, 0) for q in union]
    assert uniq == {"A": len(da), "B": len(db)} and trimmed == {"A": sum(da.values()), "B": sum(db.values())}


def _umi_dict(ctx, text, fmt, min_len, trim, umi, tmp_path, name):
    """parse_sample + collapse -> ([(insert, count)] in dictionary order, trimmed, text of <name>_umiCounts.csv or None)"""
    from mirge3_amd.collapse import parse_sample
    wd = tmp_path / name
    wd.mkdir()
    raw, n_rec = parse_sample(ctx, text, min_len, trim, umi, wd, "S")
    uniq = raw.collapse()
    cnt, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    order = np.argsort(first, kind="stable")
    got = [(seqs[i], int(cnt[i, 0])) for i in order]
    n = len(raw)
    uniq.close(); raw.close()
    f = wd / "S_umiCounts.csv"
    return got, n, n_rec, (f.read_text() if f.exists() else None)


def test_documented_umi_command_lines_known_answer(ctx, tmp_path):
    """`-a AACTGTAGGCACCATCAAT --qiagenumi -umi 0,12 -udd` and `-a illumina -umi 4,4 -udd` (the reference's documented UMI
    runs, docs/source/quick_start.md:286,306) on the ten reads printed below them (:294-298,310-314): let-7a-5p, five
    molecules each -- through mirge_reads_parse_umi on the GPU, and through the CLI."""
    from test_oracle_golden import ILLUMINA_4N_READS, LET7A, QIAGEN_READS
    from mirge3_amd.collapse import ILLUMINA_3P
    fq = lambda reads, k: "".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(r for r in reads for _ in range(k))).encode()
    trim = _ffi.MirgeTrim.make(adapter="AACTGTAGGCACCATCAAT", quality_back=10)
    got, n, n_rec, csv = _umi_dict(ctx, fq(QIAGEN_READS, 3), 1, 16, trim, _ffi.MirgeUmi.make(0, 12, qiagen=True, dedup=True), tmp_path, "q")
    assert got == [(LET7A, 5)] and n == 5 and n_rec == 15
    assert csv.splitlines()[:2] == ["UMISeq,transcriptSeq,UMICounts", f"GTTAGACCTGCA,{LET7A},3"] and len(csv.splitlines()) == 6
    got, n, _, csv = _umi_dict(ctx, fq(QIAGEN_READS, 3), 1, 16, trim, _ffi.MirgeUmi.make(0, 12, qiagen=True), tmp_path, "q2")
    assert got == [(LET7A, 15)] and n == 15 and csv is None
    trim = _ffi.MirgeTrim.make(adapter=ILLUMINA_3P, quality_back=10, count_per_modifier=False)
    got, n, _, csv = _umi_dict(ctx, fq(ILLUMINA_4N_READS, 2), 1, 16, trim, _ffi.MirgeUmi.make(4, 4, dedup=True), tmp_path, "i")
    assert got == [(LET7A, 5)] and n == 5 and csv.splitlines()[1] == f"TACACCTC,{LET7A},2"
    # the worker at HEAD counts after every modifier: the untrimmed read's 'insert' is a row too (oracle: same)
    trim = _ffi.MirgeTrim.make(adapter=ILLUMINA_3P, quality_back=10, count_per_modifier=True)
    got, n, _, _ = _umi_dict(ctx, fq(ILLUMINA_4N_READS, 2), 1, 16, trim, _ffi.MirgeUmi.make(4, 4, dedup=True), tmp_path, "i2")
    assert got[0] == (ILLUMINA_4N_READS[0][4:-4], 1) and got[1] == (LET7A, 5) and len(got) == 5 and n == 10
    # the two command lines themselves, on golden case 1's libraries (let-7a is not in them: the rows go to unmapped.csv)
    case = GoldenCase("case1_single")
    for tag, reads, extra in (("qia", QIAGEN_READS, ["-a", "AACTGTAGGCACCATCAAT", "--qiagenumi", "-umi", "0,12", "-udd"]),
                              ("ill", ILLUMINA_4N_READS, ["-a", "illumina", "-umi", "4,4", "-udd", "--trim-count", "once"])):
        p = tmp_path / f"{tag}.fastq"
        p.write_bytes(fq(reads, 2))
        _run_cli(["-s", str(p), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", tag, "-shh"] + extra)
        rows = (tmp_path / tag / "unmapped.csv").read_text().splitlines() + (tmp_path / tag / "mapped.csv").read_text().splitlines()
        assert [r for r in rows if r.startswith(LET7A + ",")][0].endswith(",5")
        rep = (tmp_path / tag / "annotation.report.csv").read_text().splitlines()[1].split(",")
        assert rep[1:4] == ["10", "5", "1"]  # Total Input Reads, Trimmed Reads (all) = molecules, Trimmed Reads (unique)
        assert len((tmp_path / tag / f"{tag}_umiCounts.csv").read_text().splitlines()) == 6


def test_umi_route_at_scale(ctx, ci_libs, tmp_path):
    """-umi 4,4 [-udd] on 3 M reads (4N layout, text of one sequence per line): the molecule counts per insert against a
    numpy count of the distinct (UMI, insert) pairs, 'Trimmed Reads (all)', the dictionary order (an insert's rank = the
    rank of its first tagged read), and the number of lines of <sample>_umiCounts.csv -- the two packings, the sort of the
    first indices and the record gather of mirge_reads_parse_umi over many tiles and several read groups."""
    from mirge3_amd.collapse import parse_sample
    rng = np.random.default_rng(17)
    tl = sorted(set(synth.make_reads(ci_libs, 30000, seed=3, pool=6000).to_list()))
    tl = [t for t in tl if "N" not in t][:5000] + ["ACGTACGTACGTAC", "ACGTTGCATGCATGCATGCATGCAACGTTGCATGCATGC"]  # one too short, one long
    tags = ["".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=8)) for _ in range(200)]
    n = 3_000_000
    ti = np.minimum((rng.pareto(1.2, size=n) * 40).astype(np.int64), len(tl) - 1)  # skewed: some inserts are very common
    gi = rng.integers(0, len(tags), size=n)
    tmpl = FlatSeqs.from_list(tl)
    tagf, tagb = FlatSeqs.from_list([t[:4] for t in tags]), FlatSeqs.from_list([t[4:] for t in tags])
    cols = [tagf.take(gi), tmpl.take(ti), tagb.take(gi)]
    text = np.frombuffer(FlatSeqs.join_columns(cols, b"\x00\x00\n"), dtype=np.uint8)
    text = text[text != 0]  # the two inner separators out: tag + insert + tag, one read per line
    # expectations from the STRINGS (two (insert, UMI) pairs can spell the same read: the templates overlap each other)
    full = cols[0].lengths + cols[1].lengths + cols[2].lengths
    keep = full - 8 >= 16
    width = int(full.max())
    mat = np.zeros((n, width), dtype=np.uint8)
    roff = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(full + 1, out=roff[1:])
    rows = np.repeat(np.arange(n, dtype=np.int64), full)
    within = np.arange(int(full.sum()), dtype=np.int64) - (roff[:-1] - np.arange(n))[rows]
    mat[rows, within] = text[text != 10]
    del rows, within
    tagged = mat.view(f"S{width}").reshape(-1)
    kept_idx = np.flatnonzero(keep)
    u_tag, u_first, u_cnt = np.unique(tagged[kept_idx], return_index=True, return_counts=True)
    u_first = kept_idx[u_first]  # first read that spells each distinct tagged string
    ins_of = np.array([t[4:-4] for t in u_tag.tolist()])
    for dedup in (True, False):
        wd = tmp_path / f"d{int(dedup)}"
        wd.mkdir()
        raw, n_rec = parse_sample(ctx, text, 16, None, _ffi.MirgeUmi.make(4, 4, dedup=dedup), wd, "S")
        assert n_rec == n
        uniq = raw.collapse()
        cnt, first = uniq.counts()
        seqs = uniq.unpack().to_list()
        order = np.argsort(first, kind="stable")
        ins, inv = np.unique(ins_of, return_inverse=True)
        want = np.zeros(len(ins), dtype=np.int64)
        np.add.at(want, inv, 1 if dedup else u_cnt)
        assert len(raw) == (len(u_tag) if dedup else int(keep.sum()))
        if dedup:
            assert sum(1 for _ in open(wd / "S_umiCounts.csv")) == len(u_tag) + 1
        assert dict(zip(seqs, cnt[:, 0].tolist())) == {q.decode(): int(c) for q, c in zip(ins.tolist(), want)}
        # dictionary order: inserts by the first appearance of any of their tagged reads
        first_read = np.full(len(ins), n, dtype=np.int64)
        np.minimum.at(first_read, inv, u_first)
        assert [seqs[i] for i in order] == [ins[k].decode() for k in np.argsort(first_read, kind="stable")]
        uniq.close(); raw.close()


def _umi_records(rng, n, adapter, f, b, qiagen):
    """FASTQ records of UMI libraries: [f nt UMI] insert [b nt UMI] adapter ... (4N layout) or insert adapter [b nt UMI]
    external adapter (Qiagen layout), from a few inserts and UMIs so that molecules repeat; adapters with errors,
    reads that end inside the adapter / the UMI, low-quality tails, inserts that contain themselves twice."""
    inserts = ["".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(14, 34)))) for _ in range(60)]
    inserts += ["ACGTACGTACGTACGTAC" + "GG" + "ACGTACGTACGTACGTAC", "A" * 20, "TGAGGTAGTAGGTTGTATAGTT"]
    umis = ["".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=f + b)) for _ in range(9)]
    recs = []
    for i in range(n):
        ins = inserts[int(rng.integers(0, len(inserts)))]
        u = umis[int(rng.integers(0, len(umis)))]
        ad = list(adapter)
        kind = int(rng.integers(0, 8))
        if kind == 1:
            ad[int(rng.integers(0, len(ad)))] = "ACGT"[int(rng.integers(0, 4))]
        elif kind == 2:
            del ad[int(rng.integers(1, len(ad) - 1))]
        ad = "".join(ad)
        ext = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC"
        seq = (ins + ad + u[f:] + ext) if qiagen else (u[:f] + ins + u[f:] + ad + ext)
        seq = seq[:int(rng.integers(30, 91))] if kind in (3, 4) else seq[:76]
        if kind == 5:
            seq = ins  # no adapter at all
        q = np.full(len(seq), ord("I"), dtype=np.uint8)
        if rng.random() < 0.25:
            k = int(rng.integers(1, 12))
            q[-k:] = rng.integers(33, 48, size=min(k, len(seq)))
        if rng.random() < 0.05:
            q[:int(rng.integers(1, 4))] = 35
        recs.append((seq, q.tobytes().decode()))
    return recs + [("ACGT", "IIII"), (adapter, "I" * len(adapter))]


@pytest.mark.parametrize("f,b,qiagen,q_front", [(4, 4, False, 0), (0, 12, True, 0), (0, 12, True, 8), (3, 0, False, 0), (0, 5, False, 0),
                                                 (2, 7, True, 0), (0, 0, True, 0), (6, 30, False, 0)])
@pytest.mark.parametrize("dedup", [False, True])
def test_umi_route_equals_the_restated_worker_and_baking(ctx, tmp_path, f, b, qiagen, q_front, dedup):
    """mirge_reads_parse_umi + mirge_collapse against the oracle's restatement of the worker's UMI branches
    (digest.py:334-365) and baking's UMI stage (:164-205): the final dictionary in order, 'Trimmed Reads (all)', and
    <sample>_umiCounts.csv byte for byte; -umi f,b with either end 0, --qiagenumi incl. its first-occurrence rule and
    b == 0, counting after every modifier (HEAD) and once."""
    adapter = "AACTGTAGGCACCATCAAT" if qiagen else "TGGAATTCTCGGGTGCCAAGGAACTCCAG"
    rng = np.random.default_rng(100 * f + b + qiagen)
    recs = _umi_records(rng, 5000, adapter, f, b, qiagen)
    text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
    for per_modifier in (True, False):
        trim = _ffi.MirgeTrim.make(adapter=adapter, quality_back=10, quality_front=q_front, count_per_modifier=per_modifier)
        got, n, n_rec, csv = _umi_dict(ctx, text, 1, 16, trim, _ffi.MirgeUmi.make(f, b, qiagen=qiagen, dedup=dedup), tmp_path,
                                       f"w{int(per_modifier)}")
        keys = oracle.umi_worker_reads(recs, dict(q_back=10, q_front=q_front, adapter=adapter), f, b, 16, qiagen, per_modifier)
        want, trimmed, rows = oracle.umi_baking(keys, f, b, 16, dedup)
        assert n_rec == len(recs) and n == trimmed
        assert got == want and len(want) > 30
        assert csv == ("".join(rows) if dedup else None)


# ---------------------------------------------------------------- size-independent properties
def test_properties_at_scale(ctx):
    """BASELINE configs C2/C3 shape (human-sized small libraries, millions of reads): invariants
    that need no oracle, plus the indexed oracle on a random sample of the collapsed reads."""
    sl = synth.make_libraries(seed=20260101, scale="small")
    casc = Cascade(ctx, sl.libs)
    reads = synth.make_reads_chunked(sl, 3_000_000, seed=11)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq = raw.collapse()
    counts, first = uniq.counts()
    assert int(counts.sum()) == len(reads)                      # collapse conserves reads
    res = casc.run(uniq)
    ps, ref, off, mm = res.fetch()
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, 0, 8, len(sl.libs["mirna"]))
    c = counts[:, 0].astype(np.int64)
    for p in range(9):                                           # a checksum of checksums
        assert cls[p, 0] == c[ps == p].sum()
    assert cls.sum() + c[ps < 0].sum() == len(reads)
    assert ex.sum() == cls[0, 0] and iso.sum() == cls[8, 0]
    useq = uniq.unpack()
    lens = useq.lengths
    assert (lens[ps == 0] < 26).all() and (lens[ps == 1] > 25).all()   # subset rules
    # idempotence / order independence: a shuffled subset gets the same per-read answers
    rng = np.random.default_rng(1)
    pick = rng.permutation(len(uniq))[:200000]
    sub = useq.take(pick)
    g = casc.annotate(sub)
    for a, b in zip(g, (ps, ref, off, mm)):
        assert np.array_equal(a, b[pick])
    # and the oracle agrees on a sample of it
    o = oracle.cascade(sub.data[: sub.offsets[20000]], sub.offsets[:20001], oracle_libs_from(sl.libs), n_pass=9,
                       indexed=True)
    for a, b in zip(o, g):
        assert np.array_equal(a.astype(np.int64), b[:20000].astype(np.int64))
    # pass-0-only run (BASELINE config C2) equals the pass-0 part of the full cascade
    c2 = Cascade(ctx, {"mirna": sl.libs["mirna"]}, n_pass=1)
    p0 = c2.annotate(sub)
    assert np.array_equal(p0[0] == 0, g[0] == 0) and np.array_equal(p0[1][p0[0] == 0], g[1][g[0] == 0])
    res.close(); uniq.close(); raw.close(); casc.close(); c2.close()


def test_bench_two_ranks_share_the_gpu(tmp_path):
    """The N > 1 code path of bench.py (rank-specific samples, barrier, max-over-ranks time, rank-0
    JSON) with two ranks on the single GPU of the test box (gloo for the barrier; the driver's real
    multi-GPU runs use RCCL, one GPU per rank)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, MIRGE_BENCH_SHARE_GPU="1", MIRGE_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--reads", "300000", "--scale", "ci", "--cpu-baseline", "0", "--pmc", "0", "--min-seconds", "0.5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "cpu_baseline" not in d and d["roofline"]["frac"] > 0
    # under torch.distributed.run without --workload the line describes BASELINE configs[3], with the numbers actually run
    assert d["config"]["workload"].startswith("C4: 2 samples x 0.3M reads") and d["config"]["raw_reads_per_gpu"] == 300000
    assert "rccl_ranks_seen" in d and d["rccl_ranks_seen"] is None  # gloo carried the barrier: RCCL was not probed
    assert d["barrier_backend"] == "gloo" and d["timing"]["timed_seconds"] >= 0.5 and d["timing"]["regions"] >= 1  # --min-seconds 0.5 below
    assert d["timing"]["timed_steps"] == d["timing"]["regions"] * 2
    # round 6: the product's route end to end (FASTQ files -> all files, tail included), both tails, the same files
    e2e = d["c4_end_to_end"]
    assert "error" not in e2e, e2e
    assert e2e["same_files_both_tails"] is True and e2e["ranges"]["wall_s"] > 0 and e2e["rank0_alone"]["wall_s"] > 0
    assert len(e2e["ranges"]["range_tail_per_rank"]) == 2 and e2e["ranges"]["output_bytes"]["mapped.csv"] > 1_000_000
    # the RCCL guard: when RCCL cannot come up (here: made to fail; on this box two ranks on one GPU would fail by themselves)
    # every rank falls back to gloo together and the line says which backend carried the barrier
    env2 = dict(os.environ, MIRGE_BENCH_SHARE_GPU="1", MIRGE_BENCH_FORCE_NCCL_FAIL="1", OMP_NUM_THREADS="4")
    cmd[cmd.index("29533")] = "29571"
    cmd[cmd.index("0.5")] = "0"
    r = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["barrier_backend"] == "gloo" and "forced" in d["barrier_backend_note"] and d["timing"]["regions"] == 1 and d["value"] > 0
    assert d["rccl_ranks_seen"] is None and d["config"]["workload"].startswith("C4:")


def test_bench_barrier_over_rccl_with_one_rank(tmp_path):
    """RCCL itself on the test box: two ranks on one GPU cannot form an RCCL communicator, one rank can.  bench.py's N > 1 route
    (gloo default group, RCCL group on top, probe all-reduce, barrier and max-over-ranks on the RCCL group) with ONE rank under
    torch.distributed.run (MIRGE_BENCH_FORCE_DIST=1).  What it pins: librccl loads and initialises in this image on this pool's
    hosts, and every collective call of the bench works on device tensors.  What it cannot: a second rank, xGMI."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, MIRGE_BENCH_FORCE_DIST="1", OMP_NUM_THREADS="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "300000",
           "--scale", "ci", "--cpu-baseline", "0", "--pmc", "0", "--min-seconds", "0", "--workload", "c3", "--cli-path", "0", "--read-sets", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "rccl_one_rank.json"), "w") as fh:
            json.dump({k: d.get(k) for k in ("barrier_backend", "barrier_backend_note", "rccl_ranks_seen", "n_gpus", "value", "ms_per_step")}, fh)
    except OSError:
        pass
    assert d["n_gpus"] == 1 and d["value"] > 0
    if d["barrier_backend"] != "nccl":  # RCCL is the pool's business, not this code's: the run fell back to gloo as designed -- say so, do not pass
        assert d["barrier_backend"] == "gloo" and d["barrier_backend_note"]
        pytest.skip("RCCL did not come up on this box: " + str(d["barrier_backend_note"])[:200])
    assert d["rccl_ranks_seen"] == 1


def test_bench_single_gpu_line(tmp_path):
    """bench.py at N = 1 on a small configuration: the ONE JSON line with the contract's fields, the oracle-checked
    sample, the roofline object of the dominant kernel and the two host-inclusive paths."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--reads", "400000", "--scale", "ci",
           "--pmc", "0", "--min-seconds", "0.5"]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="8"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["vs_baseline"] is None and d["dtype"] == "u64"
    tm = d["timing"]
    assert tm["timed_seconds"] >= 0.5 and tm["timed_steps"] == 3 * tm["regions"] and d["barrier_backend"] is None
    assert tm["step_ms_rank0"]["min"] <= tm["step_ms_rank0"]["median"] <= tm["step_ms_rank0"]["max"]
    assert abs(d["ms_per_step"] - tm["timed_seconds"] / tm["timed_steps"] * 1e3) < 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["peak"] == 8000.0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-5
    # the whole-cascade kernel is priced at 14 B per collapsed read of its group; the per-(read, pass) figure stays beside it
    if ro["kernel"].startswith("k_cascade"):
        assert ro["algorithmic_bytes_per_unit"] == 14 + 8 * (int(ro["kernel"].partition(".w")[2].rstrip("n")) - 1)
        assert 0 < ro["units_per_launch"] <= d["config"]["unique_reads_per_gpu"]
        assert abs(ro["achieved"] - ro["algorithmic_bytes_per_unit"] * ro["units_per_launch"] / (ro["avg_launch_ms"] * 1e-3) / 1e9) < 1e-2 * ro["achieved"] + 1e-3
        pp = ro["per_pass_units"]
        assert pp["units_per_launch"] >= ro["units_per_launch"] and pp["frac"] >= ro["frac"]
    assert abs(d["collapsed_reads_per_s_M"] - d["config"]["unique_reads_per_gpu"] / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * d["collapsed_reads_per_s_M"]
    assert d["raw_reads_per_s_fastq_in_M"] == d["fastq_text_path"]["M_reads_per_s"] and d["config"]["workload"].startswith("C3: 0.4M-read")
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert d["parity_on_cpu_sample"] is True and d["fastq_text_path"]["same_counts_as_step"] is True
    # round 5: every stage priced on SURVEY 8(d)'s bytes, the two other read sets of 8(d) beside `value`, how the cascade walks
    st = d["roofline_stages"]
    assert {"collapse", "cascade", "join"} <= set(st) and st["cascade"]["frac"] == ro["frac"]
    for k in ("collapse", "join"):
        assert st[k]["ms"] > 0 and abs(st[k]["achieved"] - st[k]["algorithmic_bytes"] / (st[k]["ms"] * 1e-3) / 1e9) < 1e-2 * st[k]["achieved"] + 1e-2
        assert abs(st[k]["frac"] - st[k]["achieved"] / 8000.0) < 1e-5 and st[k]["kernels"]
    U, N = d["config"]["unique_reads_per_gpu"], d["config"]["raw_reads_per_gpu"]
    assert st["join"]["algorithmic_bytes"] == 9 * U + 8 * (2 * st["join"]["units"]["mirna_references"] + 10)
    assert st["collapse"]["units"]["raw_reads"] <= N and st["collapse"]["units"]["unique_reads"] <= U
    ca = d["collapsed_reads_annotation_on_host"]  # SURVEY 8(d)'s "collapsed reads/s" as written: annotation + tables back on the host
    assert ca["unique_reads"] == U and ca["M_collapsed_reads_per_s"] > 0 and 0 < ca["annotated"] <= U
    assert abs(ca["M_collapsed_reads_per_s"] - U / (ca["ms"] * 1e-3) / 1e6) < 0.02 * ca["M_collapsed_reads_per_s"] + 0.1
    gt = d["gff_typing"]  # row N2 measured: every miRNA read of the sample through k_isotype
    assert 0 < gt["isomir_records"] <= gt["mirna_reads"] <= ca["annotated"] and gt["k_isotype_ms"] > 0 and gt["ms_call"] >= gt["k_isotype_ms"]
    assert gt["roofline"]["algorithmic_bytes"] == 350 * gt["mirna_reads"] and 0 < gt["roofline"]["frac"] < 1
    rs = d["read_sets"]
    assert rs["default_draw"]["ms_per_step"] == d["ms_per_step"] and rs["distinct"]["U_over_N"] == 1.0
    assert rs["zipf_pool"]["U_over_N"] < rs["default_draw"]["U_over_N"] < 1.0 and rs["zipf_pool"]["raw_reads"] == N
    assert d["cascade_walks"]["passes"] == 7 and d["cascade_walks"]["walks"] in (5, 7) and d["cascade_walks"]["exact_lookup_passes"] in (0, 2)


def test_variant_tally_vs_oracle(ctx):
    """Config-5 primitive (mirge_variant_tally / k_tally): membership, alignment to the family's canonical sequence,
    judgeAllign and the per-(family, sample) counts + per-position census in three variants, GPU vs the string
    restatement the reference's own functions pinned (tests/test_a2i_gff_oracle.py) -- with merged families whose
    canonical sequence is not the read's own reference, a retained mask and an RPM gate that cuts."""
    from mirge3_amd import a2i
    sl = synth.make_libraries(seed=52, scale="ci")
    rng = np.random.default_rng(6)
    mir = sl.libs["mirna"].seqs.to_list()
    hp = sl.libs["hairpin"].seqs.to_list()
    reads = synth.make_reads(sl, 12000, seed=3, mix=dict(exact=0.4, isomir=0.5, random=0.1)).to_list()
    for i in range(0, len(mir), 3):                       # planted A->G edits and end variants
        s = mir[i]
        for q, b in enumerate(s[:-5]):
            if b == "A" and rng.random() < 0.3:
                reads += [s[:q] + "G" + s[q + 1:]] * int(rng.integers(1, 6))
        h, o = int(sl.mir_hairpin[i]), int(sl.mir_hairpin_off[i])
        reads += [hp[h][max(o - 1, 0):o + len(s)], hp[h][o + 1:o + len(s) + 2], hp[h][o + 2:o + len(s)], s + "A", s[:-1] + "N"]
    reads = [r for r in reads if len(r) >= 16]
    fs = FlatSeqs.from_list(reads)
    half = len(reads) // 2
    sid = np.r_[np.zeros(half, np.int32), np.ones(len(reads) - half, np.int32)]
    casc = Cascade(ctx, sl.libs)
    raw = _ffi.DeviceReads.pack(ctx, fs)
    uniq = raw.collapse(sid, 2)
    res = casc.run(uniq)
    ps, ref, off, mm = res.fetch()
    counts, _ = uniq.counts()
    useq = uniq.unpack().to_list()
    fam_names, fam_of_ref = a2i.families(sl.libs["mirna"].names, sl.merges)
    first_member = {}
    for r, f in enumerate(fam_of_ref):
        first_member.setdefault(int(f), r)
    targets = [mir[first_member[f]] for f in range(len(fam_names))]
    retained = (rng.random(len(useq)) < 0.8).astype(np.uint8)
    freq = np.array([0.4, 1e300])  # sample 1: an isomiR read needs 3 copies; sample 2: every read is a member
    g = a2i.tally(casc, uniq, res, fam_of_ref, FlatSeqs.from_list(targets), retained, freq, per_read=True)
    o = oracle.variant_tally(useq, counts.astype(np.int64), ps, ref, fam_of_ref, targets, retained, freq)
    for k in ("diag", "state", "n_seqs", "seq_true", "count_true", "canon", "kept_exact", "census"):
        assert np.array_equal(g[k], o[k]), k
    assert g["count_true"].sum() > 5000 and g["census"][:, :, 0, 2, 2, :].sum() > 50 and (g["state"] == 0).sum() > 20
    assert (g["state"] == -1).sum() > (ps < 0).sum()  # the gate dropped isomiR reads
    # the defaults (bench.py's config 5): every miRNA its own family, every read retained and a member
    g2 = a2i.tally(casc, uniq, res, per_read=True)
    o2 = oracle.variant_tally(useq, counts.astype(np.int64), ps, ref, np.arange(len(mir)), mir)
    for k in ("diag", "state", "count_true", "canon", "census"):
        assert np.array_equal(g2[k], o2[k]), k
    res.close(); uniq.close(); raw.close(); casc.close()


def test_cli_end_to_end_single_process(tmp_path):
    """`python -m mirge3_amd.cli` on FASTQ files expanded from golden case 2: the per-miRNA tables must be
    the reference's byte for byte; the report differs only in 'Total Input Reads' (the golden counters
    pretend 17 reads were lost to trimming)."""
    import csv as _csv
    import subprocess
    import sys
    case = GoldenCase("case2_two_samples")
    files = []
    for s, nm in enumerate(case.samples):
        p = tmp_path / f"{nm}.fastq"
        with open(p, "w") as fh:
            k = 0
            for seq, row in zip(case.seqs, case.counts):
                for _ in range(int(row[s])):
                    fh.write(f"@r{k}\n{seq}\n+\n{'I' * len(seq)}\n")
                    k += 1
            fh.write("@short\nACGTACGT\n+\nIIIIIIII\n")  # below --minimum-length: counted as input only
        files.append(str(p))
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, "-c", "import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()" % root,
           "-s", ",".join(files), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "out", "-shh"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = tmp_path / "out"
    for f in ("miR.Counts.csv", "miR.RPM.csv"):
        assert (out / f).read_text() == case.text(f), f
    got = list(_csv.DictReader(open(out / "annotation.report.csv")))
    exp = list(_csv.DictReader(open(os.path.join(case.dir, "annotation.report.csv"))))
    for g_row, e_row in zip(got, exp):
        for k in e_row:
            if k == "Total Input Reads":
                assert int(g_row[k]) == int(e_row["Trimmed Reads (all)"]) + 1
            else:
                assert g_row[k] == e_row[k], k
    # mapped.csv: same rows as the reference's (the outer join sorts the sequences)
    assert (out / "mapped.csv").read_text() == case.text("mapped.csv")
    assert (out / "unmapped.csv").read_text() == case.text("unmapped.csv")
    assert "Alignment completed" in (out / "run.log").read_text()
    # save (-spl) and resume (-rr) through the reference's two pickle files (mirge/__main__.py:91-108,142-148)
    r = subprocess.run(cmd[:-3] + ["-dn", "out_spl", "-shh", "-spl"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    spl = tmp_path / "out_spl"
    assert (spl / "collapsed.pkl").exists() and (spl / "collapsed_accessories.pkl").exists()
    cmd_rr = cmd[:3] + ["-s", str(spl)] + cmd[5:-3] + ["-dn", "out_rr", "-shh", "-rr"]
    r = subprocess.run(cmd_rr, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for f in ("miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv", "annotation.report.csv"):
        assert (tmp_path / "out_rr" / f).read_text() == (out / f).read_text(), f
    with pytest.raises(SystemExit):
        from mirge3_amd.cli import parse_args
        parse_args(["-s", "x.fastq", "-lib", "L", "-on", "human", "-nmir"])


def _case_fastqs(case, tmp_path):
    files = []
    for s, nm in enumerate(case.samples):
        p = tmp_path / f"{nm}.fastq"
        with open(p, "w") as fh:
            k = 0
            for seq, row in zip(case.seqs, case.counts):
                for _ in range(int(row[s])):
                    fh.write(f"@r{k}\n{seq}\n+\n{'I' * len(seq)}\n")
                    k += 1
        files.append(str(p))
    return files


def _run_cli(argv, timeout=600):
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, "-c", "import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()" % root]
    r = subprocess.run(cmd + list(argv), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return r


@pytest.mark.parametrize("name", ["case1_single", "case3_spikein"])
def test_cli_on_a_library_directory_of_ebwt_indexes_only(name, tmp_path):
    """What a miRge3.0 user has: a library directory that holds bowtie indexes and nothing else (row N3).  The golden
    libraries are written as .ebwt files (tests/ebwt_writer.py), every .fa is removed, and the CLI must produce the
    reference's files byte for byte."""
    import shutil
    from ebwt_writer import fasta_dir_to_ebwt
    case = GoldenCase(name)
    libdir = tmp_path / "Libs"
    shutil.copytree(case.libdir, libdir)
    fasta_dir_to_ebwt(str(libdir / ORG / "index.Libs"))
    assert not [f for f in os.listdir(libdir / ORG / "index.Libs") if f.endswith(".fa")]
    files = _case_fastqs(case, tmp_path)
    _run_cli(["-s", ",".join(files), "-lib", str(libdir), "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "out", "-shh"]
             + (["-spk"] if case.spike else []))
    out = tmp_path / "out"
    for f in ("miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv"):
        assert (out / f).read_text() == case.text(f), f


@pytest.mark.parametrize("name", ["case1_single", "case3_spikein"])
def test_cli_backend_bowtie_crosses_the_process_boundary(name, tmp_path):
    """`--backend bowtie` (BASELINE.md 3.5's C1 through this build's CLI; SURVEY section 5 'Config / flags'): collapse on the GPU,
    then the reference's ten bowtie runs verbatim across its process boundary -- here against the stand-in bowtie of
    tests/golden/fake_bowtie (no bowtie 1.x in the image or on the pool: with a real one in -pbwt this is the parity switch) --,
    then the count tables.  Every file must equal what the reference wrote for the golden case, and what `--backend gpu` writes."""
    case = GoldenCase(name)
    files = _case_fastqs(case, tmp_path)
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fake_bowtie")
    base = ["-s", ",".join(files), "-lib", str(case.libdir), "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-shh"] + (["-spk"] if case.spike else [])
    _run_cli(base + ["-dn", "bt", "--backend", "bowtie", "-pbwt", fake, "-cpu", "2"])
    _run_cli(base + ["-dn", "gpu"])
    for f in ("miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv", "annotation.report.csv"):
        if f != "annotation.report.csv":  # (its 'Total Input Reads' counts records the golden run dropped before these FASTQ files were made)
            assert (tmp_path / "bt" / f).read_text() == case.text(f), f
        assert (tmp_path / "bt" / f).read_text() == (tmp_path / "gpu" / f).read_text(), f
    assert "bowtie backend" in (tmp_path / "bt" / "run.log").read_text()


def test_integration_md_stub_runs(tmp_path):
    """The ctypes stub printed in INTEGRATION.md section 2 is executed as written (only the library path and the
    `nine_libraries` input are supplied) on golden case 1 and must reproduce the reference's annotation."""
    import re
    import pandas as pd
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = next(b for b in blocks if b.startswith("import ctypes as C, numpy as np"))
    stub = stub.replace('C.CDLL("libmirge_native.so")', "C.CDLL(%r)" % _ffi.SO_PATH)
    case = GoldenCase("case1_single")
    from helpers import PASS_LIBKEY
    nine = [(case.libs[PASS_LIBKEY[p]].names, case.libs[PASS_LIBKEY[p]].seqs.to_list()) for p in range(9)]
    ns = {"nine_libraries": nine}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    df = pd.DataFrame(case.counts, columns=case.samples, index=pd.Index(case.seqs, name="Sequence"))
    df = df.assign(**dict.fromkeys(PASS_COLS, ''))
    df = df.assign(annotFlag=0).reindex(columns=['annotFlag'] + PASS_COLS + case.samples)
    out = ns["bwtAlign"](None, df, str(tmp_path), DB)
    out[out.annotFlag.eq(1)].to_csv(tmp_path / "mapped.csv")
    assert (tmp_path / "mapped.csv").read_text() == case.text("mapped.csv")


@pytest.mark.parametrize("n", [65535, 65536, 65537, 131073, 300001, 2500000])
@pytest.mark.parametrize("shape", ["uniform", "zipf", "one_third", "burst"])
def test_partitioned_collapse_sizes_and_skew(ctx, n, shape):
    """The key path of collapse around its thresholds (global-atomic table below 65536 reads, one radix level up to
    64 buckets, two levels beyond) on duplicates that are uniform, Zipf-distributed, one sequence in every third read, or
    bursts of one sequence behind distinct reads (the chunk cache of k_part_agg, the append regions -- overflowing in the
    last case -- and the exact second level all see their worst case): the
    dictionary -- sequence -> (count, first index) -- must equal numpy's."""
    rng = np.random.default_rng(n % 1000 + len(shape))
    n_tmpl = max(n // 3, 1000)
    lens = rng.integers(16, 32, size=n_tmpl)
    off = np.zeros(n_tmpl + 1, np.int64); np.cumsum(lens, out=off[1:])
    tmpl = FlatSeqs(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(off[-1]))], off)
    if shape == "uniform":
        idx = rng.integers(0, n_tmpl, size=n)
    elif shape == "zipf":
        idx = np.minimum(rng.zipf(1.2, size=n) - 1, n_tmpl - 1)
    elif shape == "one_third":
        idx = np.where(rng.random(n) < 0.33, 7, rng.integers(0, n_tmpl, size=n))
    else:  # every writer's chunk starts with distinct reads (its cache fills) and ends with a burst of ONE sequence: the
        # burst's copies are turned away by the cache and merged per wave (k_part_agg); without that merge they overflow one
        # level-1 region and the call is redone with chunk-sized regions (collapse_phase_a, attempt 1)
        idx = rng.integers(0, n_tmpl, size=n)
        chunk = max(n // 256, 2048)
        pos = np.arange(n) % chunk
        idx[pos >= chunk // 2] = 11
    reads = tmpl.take(idx)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    u = raw.collapse()
    cnt, first = u.counts()
    got_seqs = u.unpack().to_list()
    # templates may coincide by chance: the expectation goes through the sequences, not the template ids
    tl = tmpl.to_list()
    canon = {}
    key_of = np.empty(n_tmpl, np.int64)
    for t, sq in enumerate(tl):
        key_of[t] = canon.setdefault(sq, t)
    keys = key_of[idx]
    uk, ufirst, ucnt = np.unique(keys, return_index=True, return_counts=True)
    want = {tl[k]: (int(c), int(f)) for k, f, c in zip(uk, ufirst, ucnt)}
    got = {sq: (int(c), int(f)) for sq, c, f in zip(got_seqs, cnt[:, 0], first)}
    assert len(got_seqs) == len(want) and got == want
    u.close(); raw.close()


@pytest.mark.parametrize("hook", ["MIRGE_TEST_SMALL_PART", "MIRGE_TEST_SMALL_REGION", "MIRGE_TEST_SMALL_REGION+MIRGE_TEST_PART_OOM"])
def test_collapse_partition_overflow_falls_back(tmp_path, hook):
    """A bucket holding more distinct reads than its LDS table raises the overflow flag and the call is redone
    with the global-atomic tables: forced here with the MIRGE_TEST_SMALL_PART hook in a fresh process.  The same flag is
    raised when a level-1 region of the radix split would overflow its fixed capacity (MIRGE_TEST_SMALL_REGION halves them)."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import mirge3_amd
from mirge3_amd import _ffi
from mirge3_amd.seqio import FlatSeqs
rng = np.random.default_rng(5)
n = 400000
lens = rng.integers(16, 31, size=n)
off = np.zeros(n + 1, np.int64); np.cumsum(lens, out=off[1:])
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(off[-1]))]
reads = FlatSeqs(data, off)
dup = reads.take(rng.integers(0, n, size=100000))
allr = FlatSeqs(np.concatenate([reads.data, dup.data]), np.concatenate([reads.offsets, dup.offsets[1:] + reads.offsets[-1]]))
ctx = _ffi.Context(0)
raw = _ffi.DeviceReads.pack(ctx, allr)
u = raw.collapse()
cnt, first = u.counts()
from collections import Counter
exp = Counter(allr.to_list())
got = dict(zip(u.unpack().to_list(), cnt[:, 0].tolist()))
assert got == dict(exp), (len(got), len(exp))
print("OK", len(got))
""" % root
    # the third form: the roomy second attempt (2 KiB of HBM per read) cannot get its memory (MIRGE_TEST_PART_OOM makes its
    # first allocation fail as an exhausted device would) -- the call must go on to the global-atomic tables, not fail
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{h: "1" for h in hook.split("+")}),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_collapse_count_matrix_beyond_2_to_32_cells(ctx):
    """The general path's slot table x sample matrix passes 2^32 cells from ~24 samples x 4 M reads on (here: 2^20 slots
    x 4100 samples = 17 GB, sized for 288 GB of HBM): every cell must start at zero -- the init kernel walks it in 64
    bits -- or the per-sample counts come out of recycled pool memory."""
    rng = np.random.default_rng(9)
    S, n, T = 4100, 300_000, 500
    tl = ["".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(L))) for L in rng.integers(16, 40, size=T)]
    tmpl = FlatSeqs.from_list(tl)
    # dirty the pool first: a block of the matrix's size, filled with ones, goes back to the pool
    import ctypes as C
    pick = rng.integers(0, T, size=n)
    sid = rng.integers(0, S, size=n).astype(np.int32)
    sid[:S] = np.arange(S)  # every sample occurs
    reads = tmpl.take(pick)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    for _ in range(2):  # the second call reuses the first call's (now non-zero) matrix block
        uniq = raw.collapse(sid, S)
        cnt, first = uniq.counts()
        seqs = uniq.unpack().to_list()
        assert sorted(seqs) == sorted(set(tl[k] for k in pick)) and cnt.shape == (len(seqs), S)
        want = np.zeros((T, S), dtype=np.int64)
        np.add.at(want, (pick, sid), 1)
        by_seq = {}
        for k in range(T):
            by_seq[tl[k]] = by_seq.get(tl[k], 0) + want[k]
        for i, q in enumerate(seqs):
            assert np.array_equal(cnt[i].astype(np.int64), by_seq[q]), q
        assert int(cnt.sum()) == n
        uniq.close()
    raw.close()


def test_exact_passes_are_whole_read_lookups(ctx, ci_libs, ci_cascade):
    """Round 5: the passes that admit no mismatch anywhere in the read -- pass 0 ("-n 0", len < 26: inside the seed) and pass 3
    ("-v 0" on the read without its T tail), manifoldAlign.py:85,93,118-126 -- are answered by ONE lookup of the whole read in
    a table of every valid window of the library, riding in the walk of pass 1 / pass 2 (kernels_cascade.hpp, ExactStep).
    Against the oracle on reads made for it: every substring of miRNAs at lengths 12-25 (a hit at every offset; SNP variants
    share substrings, so the LOWEST position must win), the same with one substitution or an N (no exact hit: later passes),
    25/26-nt reads either side of the length rule, pre-tRNA tails with heads of every length 1-28 plus 3-7 Ts, heads that are
    no substring, all-T reads (empty head), reads of 32+ nt beside them (other read groups, other kernels)."""
    rng = np.random.default_rng(11)
    mir, pre, hp = ci_libs.libs["mirna"], ci_libs.libs["pre_trna"], ci_libs.libs["hairpin"]
    reads = []
    for r in range(0, len(mir), 7):
        s = mir.seqs.get(r)
        for L in range(12, len(s) + 1):
            for o in range(len(s) - L + 1):
                reads.append(s[o:o + L])
                if (L + o) % 5 == 0:
                    q = int(rng.integers(0, L))
                    x = list(s[o:o + L]); x[q] = "ACGT"[("ACGT".index(x[q]) + 1 + int(rng.integers(0, 3))) % 4]
                    reads.append("".join(x))
                if (L + o) % 11 == 0:
                    x = list(s[o:o + L]); x[int(rng.integers(0, L))] = "N"
                    reads.append("".join(x))
    for r in range(0, len(hp), 9):  # 25-31-nt pieces of hairpins: some contain a whole miRNA (longer than the rule allows)
        s = hp.seqs.get(r)
        for L in (24, 25, 26, 27, 31):
            o = int(rng.integers(0, len(s) - L + 1))
            reads.append(s[o:o + L])
    for r in range(0, len(pre), 3):
        s = pre.seqs.get(r)
        for hl in range(1, 29):
            for tails in (3, 4, 7):
                if hl + tails > 31:
                    continue
                o = int(rng.integers(0, len(s) - hl + 1))
                head = s[o:o + hl] if (hl + tails) % 6 else s[len(s) - hl:]
                reads.append(head + "T" * tails)
                if hl > 4 and (hl + r) % 4 == 0:  # a head one base off: no exact window
                    x = list(head); x[hl // 2] = "ACGT"[("ACGT".index(x[hl // 2]) + 1) % 4]
                    reads.append("".join(x) + "T" * tails)
    reads += ["TTT", "TTTT", "T" * 31, "A", "ACGTTT", "NTTTT", "TTTA"]
    reads += [hp.seqs.get(3)[:40], pre.seqs.get(2)[10:60] + "TTTT", mir.seqs.get(1) + "ACGTACGTACGTACG"]
    reads = sorted(set(reads))
    rng.shuffle(reads)
    fs = FlatSeqs.from_list(reads)
    g = ci_cascade.annotate(fs)
    walks = _ffi.cascade_walks(ctx)
    # exact miRNA | hairpin, mature tRNA | primary tRNA, snoRNA+rRNA+ncRNA (one merged pass), mRNA, isomiR
    riding = {(5, 2, 7), (7, 2, 7)}  # (7: a build whose exact steps walk on their own, MIRGE_EXACT_RIDE=0)
    assert walks in ({(7, 0, 7)} if os.environ.get("MIRGE_EXACT_WALKS") == "0" else riding), walks
    o = oracle.cascade(fs.data, fs.offsets, oracle_libs_from(ci_libs.libs), n_pass=9)
    _assert_same(o, g)
    assert (g[0] == 0).sum() > 500 and (g[0] == 3).sum() > 100 and (g[0] == 8).sum() > 50 and (g[0] == 1).sum() > 5
    # the same walks in the build WITH N masks: a set in which most reads hold an ambiguous call makes that group the bulk group
    # (k_cascade_bulk<1, true>).  An N is a mismatch wherever it is aligned: no exact pass may take such a read, later passes may
    withn = []
    for q in reads:
        if len(q) > 3 and rng.random() < 0.75:
            j = int(rng.integers(0, len(q)))
            q = q[:j] + "N" + q[j + 1:]
        withn.append(q)
    fs = FlatSeqs.from_list(sorted(set(withn)))
    g = ci_cascade.annotate(fs)
    o = oracle.cascade(fs.data, fs.offsets, oracle_libs_from(ci_libs.libs), n_pass=9)
    _assert_same(o, g)
    has_n = np.array(["N" in q for q in fs.to_list()])
    assert has_n.sum() > 2 * (~has_n).sum() and not ((g[0] == 0) & has_n).any() and ((g[0] >= 1) & has_n).sum() > 100


def test_whole_read_tables_are_dropped_and_rebuilt(ctx, ci_libs, ci_cascade):
    """A library keeps the whole-read tables of its last few read-length sets only (lib_exact_trim: beyond six everything is
    dropped behind a synchronisation and the walk lists rebuilt).  Ten batches of ten different length sets through one
    cascade, the first one again at the end: every batch equals the oracle."""
    mir, pre = ci_libs.libs["mirna"], ci_libs.libs["pre_trna"]
    olibs = oracle_libs_from(ci_libs.libs)
    batches = []
    for k in range(10):
        L = 16 + k % 8
        reads = []
        for r in range(k, len(mir), 11):
            s = mir.seqs.get(r)
            if len(s) >= L:
                reads.append(s[:L])
                reads.append(s[len(s) - L:])
            p_ = pre.seqs.get(r % len(pre))
            reads.append(p_[len(p_) - (L - 4 - k // 8):] + "TTTT")     # a pre-tRNA tail: pass 3 at a head length of its own
        reads.append("ACGT" * 4 + "A" * (k + 1))                          # (makes every batch's length set differ)
        batches.append(FlatSeqs.from_list(sorted(set(reads))))
    for fs in batches + batches[:1]:
        g = ci_cascade.annotate(fs)
        o = oracle.cascade(fs.data, fs.offsets, olibs, n_pass=9)
        _assert_same(o, g)
        assert (g[0] == 0).sum() > 10 and (g[0] == 3).sum() > 5


def test_a_borrowed_library_keeps_the_probe_path(ctx, ci_libs, ci_cascade):
    """A library made in one context and handed to another context of the same device (allowed: `cascade_prepare` only asks
    for the same device): the borrower gets no whole-read tables -- only the owner builds, names and drops them, so a walk list
    of one context can never name a table another context freed -- and its annotation equals the owner's and the oracle's."""
    mir = ci_libs.libs["mirna"]
    reads = sorted({mir.seqs.get(r)[o:o + L] for r in range(0, len(mir), 5) for L in (17, 21, 22) for o in (0, 1)
                    if len(mir.seqs.get(r)) >= o + L})
    fs = FlatSeqs.from_list(reads)
    own = ci_cascade.annotate(fs)
    assert _ffi.cascade_walks(ctx)[1] == (0 if os.environ.get("MIRGE_EXACT_WALKS") == "0" else 2)
    ctx2 = _ffi.Context(0)
    try:
        dr = _ffi.DeviceReads.pack(ctx2, fs)
        res = _ffi.cascade_run(ctx2, dr, ci_cascade.dev_libs, ci_cascade.policies, ci_cascade._prepared)
        got = res.fetch()
        assert _ffi.cascade_walks(ctx2) == (7, 0, 7)
        res.close(); dr.close()
    finally:
        ctx2.close()
    _assert_same(own, got)
    _assert_same(oracle.cascade(fs.data, fs.offsets, oracle_libs_from(ci_libs.libs), n_pass=9), got)
    assert (got[0] == 0).sum() > 20


def test_long_reads_under_a_length_rule_beyond_255(ctx, ci_libs):
    """A policy whose `len >` rule lies beyond 255 (round 4's review): the long class stands for every length from 256 on, so
    its tables must be built whatever the rule says -- the 400-nt read passes `len > 300` on the device and used to probe a
    table nobody had built."""
    mrna = ci_libs.libs["mrna"]
    lib = _ffi.DeviceLibrary(ctx, mrna.seqs)
    ref = mrna.seqs.get(4)
    reads = [ref[20:420], ref[5:280], ref[100:130], ref[7:263]]
    pol = _ffi.MirgePolicy()
    for k, v in dict(mode=0, mm=1, seedlen=28, maxtotal=2, len_gt=300).items():
        setattr(pol, k, v)
    pol2 = _ffi.MirgePolicy()
    for k, v in dict(mode=1, mm=0, seedlen=28, maxtotal=0, len_lt=270).items():
        setattr(pol2, k, v)
    dr = _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list(reads))
    res = _ffi.cascade_run(ctx, dr, [lib, lib], [pol, pol2])
    ps, rf, of, mm = res.fetch()
    # 400 nt: pass 0 (len > 300); 275 nt: neither rule; 30 nt and 256 nt: pass 1 (len < 270, exact)
    assert ps.tolist() == [0, -1, 1, 1] and rf.tolist() == [4, -1, 4, 4] and of.tolist() == [20, -1, 100, 7] and mm.tolist() == [0, -1, 0, 0]
    res.close(); dr.close(); lib.close()


@pytest.mark.parametrize("hooks", [dict(MIRGE_FUSED_MAX="0"), dict(MIRGE_FUSED_MAX="0", MIRGE_BULK_FUSED="0"), dict(MIRGE_EXACT_WALKS="0"),
                                   dict(MIRGE_SPEC_MAX="0"), dict(MIRGE_SPEC_MAX="100000000", MIRGE_CASCADE_REP="1"),
                                   dict(MIRGE_SPEC_MAX="100000000", MIRGE_SPEC_TICKETS="1", MIRGE_XAUX_SLOTS="3"),
                                   dict(MIRGE_SCATTER_ON_XAUX="2", MIRGE_FUSED_MAX="2000", MIRGE_STREAM_PRIORITY="1")])
def test_staged_cascade_for_every_group(hooks):
    """Small read groups normally take k_cascade_fused (one launch for the whole cascade, no compaction); MIRGE_FUSED_MAX=0
    sends every group through the staged form with survivor lists instead -- k_cascade_bulk (all passes in one launch), or
    with MIRGE_BULK_FUSED=0 one k_pass launch per pass.  All must agree with the oracle: the oracle parity tests of this
    file, the one-call route (full cascade and the one-pass C2 cascade) and the cascade fuzz are re-run in a fresh process with
    the hooks set.  With MIRGE_FUSED_MAX=0 the small groups of one-word reads WITH ambiguous calls also walk with the exact
    steps of round 5 (the build with N masks); MIRGE_EXACT_WALKS=0 is round 4's one walk per pass, no whole-read tables.
    Round 6: tiny groups take k_cascade_spec + k_cascade_pick (all passes at once, the first answer per read) -- MIRGE_SPEC_MAX=0
    sends them through k_cascade_fused as before, a huge MIRGE_SPEC_MAX every small group through the speculative form, here
    together with the repeat-aware build of the kernels (MIRGE_CASCADE_REP=1) that uniform libraries do not launch by themselves;
    MIRGE_SPEC_TICKETS=1 is the measured-and-left-off form with the pick inside k_cascade_spec (the workgroup that ends a round's
    last step), MIRGE_XAUX_SLOTS=3 the small groups over three extra streams instead of two.  The last set: groups beyond 2 000 reads
    staged (so staged and one-launch small groups meet in one sample), the small groups' scatter kernels side by side regardless
    (MIRGE_SCATTER_ON_XAUX=2: by itself only without a staged small group), streams with priorities (measured, off)."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"),
                        os.path.join(root, "tests", "test_gpu_fuzz.py"), "-k",
                        "vs_oracle or vs_bruteforce or low_complexity or edge_cases or golden_cascade or random_cascade or one_call or exact_passes"],
                       env=dict(os.environ, MIRGE_FUZZ_SEEDS=os.environ.get("MIRGE_HOOK_FUZZ_SEEDS", "3"), **hooks), capture_output=True, text=True,
                       timeout=1500, cwd=root)  # (three seeds of each cascade fuzzer per alternate path: the suite's run time; the default path runs 16)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_trimming_general_instance_at_both_heights():
    """k_trim's general instance (two adapters, -n, --no-indels, read wildcards, anchored and linked adapters) is compiled for
    adapters of up to 32 and up to 64 bases; the launcher picks by adapter length.  MIRGE_TRIM_TALL=1 sends every such call
    through the 64-row instance: the linked / anchored tests of this file again, in a fresh
    process (the default run above covers the 32-row instance with the same tests, and the 64-row one with its long adapters)."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"), "-k",
                        "linked_and_anchored"],
                       env=dict(os.environ, MIRGE_TRIM_TALL="1"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_partitioned_collapse_with_sharded_output():
    """k_part_dedup can write its buckets' unique reads through eight cursors into a staging area that k_part_compact makes dense
    (chosen by itself when a context's previous sample had few unique reads); MIRGE_DEDUP_SHARDED=1 makes every partitioned
    collapse take that route: the collapse tests of this file again, in a fresh process."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"), "-k",
                        "collapse_vs_oracle or partitioned_collapse_sizes or one_call_equals"],
                       env=dict(os.environ, MIRGE_DEDUP_SHARDED="1"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_dedup_output_mode_is_chosen_on_the_device(ctx):
    """Round 6: k_part_dedup chooses between the one output cursor and the eight sharded ones ON THE DEVICE, per sample, from the
    record count k_part_agg leaves (until round 5 the host chose from the context's previous sample).  A Zipf sample (few unique
    reads: few records) and an all-distinct one through ONE context in alternation: the same dictionaries as the oracle's either way,
    and k_part_compact's launch record shows the mode followed the sample, not the sample before it."""
    sl = synth.make_libraries(seed=5, scale="ci")
    zipf = synth.make_reads(sl, 400000, seed=21, pool=20000)
    flat = synth.make_reads(sl, 400000, seed=22, mix=dict(synth.DEFAULT_MIX, exact=0.0, isomir=0.0))
    for k, reads in enumerate((zipf, flat, zipf, flat)):
        raw = _ffi.DeviceReads.pack(ctx, reads)
        u = raw.collapse()
        cnt, first = u.counts()
        o_first, o_cnt, _ = oracle.collapse(reads.data, reads.offsets)
        order = np.argsort(first, kind="stable")
        assert np.array_equal(first[order], o_first) and np.array_equal(cnt[order, 0], o_cnt), k
        u.close(); raw.close()


def test_reads_with_outlier_buckets_take_the_heavy_kernel():
    """Round 6: a read one of whose probe buckets holds more windows than MIRGE_BIG_T is not aligned by its wave (align_hybrid answers
    MIRGE_DEFER) but by k_cascade_heavy, a workgroup per read.  With the threshold at 64 windows most repeat-derived reads of the
    repeat-rich CI libraries -- and plenty of ordinary ones -- take that route: the oracle tests of this file again in a fresh process,
    every field of every read, through the one-call route, the staged one and the small groups' one-launch cascades."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py"), "-k",
                        "repeat_rich or low_complexity or golden_cascade or cascade_vs_oracle or staged"],
                       env=dict(os.environ, MIRGE_BIG_T="64"), capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    code = """
import sys; sys.path.insert(0, %r)
import numpy as np, mirge3_amd
from mirge3_amd import _ffi, synth
from mirge3_amd.cascade import Cascade
ctx = _ffi.Context(0)
sl = synth.make_libraries(seed=123, scale="ci", repeats=True)
casc = Cascade(ctx, sl.libs)
raw = _ffi.DeviceReads.pack(ctx, synth.make_reads(sl, 30000, seed=11, mix=synth.REPEAT_MIX))
ctx.profile(True); ctx.profile_reset()
u, r = casc.collapse_and_run(raw)
ctx.sync()
names = [n for n, l, ms, un in ctx.profile_records() if l]
assert any(n.startswith("k_cascade_heavy") for n in names), names
assert (r.fetch()[0] >= -1).all()  # no read is left with the 'deferred' mark
print("OK")
""" % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MIRGE_BIG_T="64"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_collapse_cascade_one_call_equals_two_calls(ctx):
    """mirge_collapse_cascade (the bulk group's passes queued behind the collapse kernels, read count taken from
    device memory) against mirge_collapse + mirge_cascade_run on the same reads: same unique reads, counts, first
    indices and per-read annotation.  Sizes on both sides of the partitioned-path threshold, and the overflow
    fallback (MIRGE_TEST_SMALL_PART) in a fresh process."""
    import subprocess
    import sys
    sl = synth.make_libraries(seed=20260101, scale="small")
    casc = Cascade(ctx, sl.libs)
    casc2 = Cascade(ctx, {"mirna": sl.libs["mirna"]}, n_pass=1)

    def canon(uniq, res):
        cnt, first = uniq.counts()
        order = np.argsort(first, kind="stable")
        seqs = uniq.unpack().to_list()
        ann = res.fetch()
        return [seqs[i] for i in order], cnt[order], first[order], [a[order] for a in ann]

    for n, seed in ((30000, 3), (400000, 4), (1500000, 5)):
        reads = synth.make_reads_chunked(sl, n, seed=seed) if n > 100000 else synth.make_reads(sl, n, seed=seed, n_frac=0.01)
        raw = _ffi.DeviceReads.pack(ctx, reads)
        u1 = raw.collapse(); r1 = casc.run(u1)
        u2, r2 = casc.collapse_and_run(raw)
        a, b = canon(u1, r1), canon(u2, r2)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
        c1 = _ffi.count_join(ctx, u1, r1, 0, 8, len(sl.libs["mirna"]))
        c2 = _ffi.count_join(ctx, u2, r2, 0, 8, len(sl.libs["mirna"]))
        assert all(np.array_equal(x, y) for x, y in zip(c1, c2))
        # BASELINE configs[1] (`bench.py --workload c2`): ONE pass through the one-call route -- the only user of k_pass with the
        # read count still on the device (a single step is never walked by k_cascade_bulk) -- against the two-call sequence,
        # against pass 0 of the full cascade, and against the oracle's pass 0 on the first reads (manifoldAlign.py:85,93)
        u3, r3 = casc2.collapse_and_run(raw)
        u4 = raw.collapse(); r4 = casc2.run(u4)
        c, d = canon(u3, r3), canon(u4, r4)
        assert c[0] == d[0] == a[0] and np.array_equal(c[1], d[1]) and np.array_equal(c[2], d[2]) and np.array_equal(c[1], a[1])
        assert all(np.array_equal(x, y) for x, y in zip(c[3], d[3]))
        hit0 = a[3][0] == 0
        assert np.array_equal(c[3][0] == 0, hit0) and ((c[3][0] == 0) | (c[3][0] == -1)).all()
        for f in (1, 2, 3):
            assert np.array_equal(c[3][f][hit0], a[3][f][hit0])
        k = min(len(c[0]), 20000)
        fs = FlatSeqs.from_list(c[0][:k])
        o = oracle.cascade(fs.data, fs.offsets, oracle_libs_from(sl.libs)[:1], n_pass=1, indexed=True)
        for x, y in zip(o, c[3]):
            assert np.array_equal(x.astype(np.int64), y[:k].astype(np.int64))
        j3 = _ffi.count_join(ctx, u3, r3, 0, -2, len(sl.libs["mirna"]))
        j4 = _ffi.count_join(ctx, u4, r4, 0, -2, len(sl.libs["mirna"]))
        assert all(np.array_equal(x, y) for x, y in zip(j3, j4)) and np.array_equal(j3[1], c1[1])
        for h in (r1, r2, r3, r4, u1, u2, u3, u4, raw):
            h.close()
    casc.close(); casc2.close()
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import mirge3_amd
from mirge3_amd import _ffi, synth
from mirge3_amd.cascade import Cascade
sl = synth.make_libraries(seed=5, scale="ci")
ctx = _ffi.Context(0)
casc = Cascade(ctx, sl.libs)
reads = synth.make_reads_chunked(sl, 600000, seed=2)
raw = _ffi.DeviceReads.pack(ctx, reads)
u1 = raw.collapse(); r1 = casc.run(u1)
u2, r2 = casc.collapse_and_run(raw)   # 64 buckets: the partition overflows, the queued cascade is discarded
def canon(u, r):
    cnt, first = u.counts(); o = np.argsort(first, kind="stable"); s = u.unpack().to_list()
    return [s[i] for i in o], cnt[o], [a[o] for a in r.fetch()]
a, b = canon(u1, r1), canon(u2, r2)
assert a[0] == b[0] and np.array_equal(a[1], b[1]) and all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))
print("OK", len(a[0]))
""" % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MIRGE_TEST_SMALL_PART="1"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def _run_cli_two_ranks(tmp_path, files, case, extra, port):
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    launcher = tmp_path / "run_cli.py"
    launcher.write_text("import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()\n" % root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(launcher), "-s", ",".join(files), "-lib", case.libdir, "-on", ORG, "-db", "miRBase",
           "-o", str(tmp_path), "-dn", "out", "-shh"] + extra
    r = subprocess.run(cmd, env=dict(os.environ, MIRGE_SHARE_GPU="1", OMP_NUM_THREADS="2"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return tmp_path / "out"


def _same_report_but_total_input(out, case):
    """annotation.report.csv against the golden file, but for 'Total Input Reads': the FASTQ files written from the golden
    dictionaries hold the kept reads only, the reference's run also saw the few shorter than --minimum-length"""
    got = (out / "annotation.report.csv").read_text().splitlines()
    exp = case.text("annotation.report.csv").splitlines()
    assert got[0] == exp[0] and len(got) == len(exp)
    for g_line, e_line in zip(got[1:], exp[1:]):
        assert g_line.split(",")[0] == e_line.split(",")[0] and g_line.split(",")[2:] == e_line.split(",")[2:]


def _case_fastq_files(case, tmp_path):
    files = []
    for s, nm in enumerate(case.samples):
        p = tmp_path / f"{nm}.fastq"
        with open(p, "w") as fh:
            for seq, row in zip(case.seqs, case.counts):
                if row[s]:
                    fh.write(f"@r\n{seq}\n+\n{'I' * len(seq)}\n" * int(row[s]))
        files.append(str(p))
    return files


def test_cli_two_ranks_one_sample_each(tmp_path):
    """The sharded CLI (torch.distributed.run, one sample per rank; rank 0 gathers the per-sample count columns and
    dictionaries over gloo, merges the dictionaries into the sample matrix on its GPU and writes the run's files): two
    ranks on the single GPU of the test box, golden case 2 -- the same bytes as the one-process run, ONE mapped.csv /
    unmapped.csv over the outer-joined frame (mirge/__main__.py:164-173, digest.py:243), -ie included."""
    case = GoldenCase("case2_two_samples")
    out = _run_cli_two_ranks(tmp_path, _case_fastq_files(case, tmp_path), case, ["-ie"], 29541)
    for f in ("miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv", "isomirs.csv", "isomirs.samples.csv"):
        assert (out / f).read_text() == case.text(f), f
    _same_report_but_total_input(out, case)
    assert not (out / "mapped.S1.csv").exists()


def test_cli_two_ranks_gff_a2i_equal_the_reference_files(tmp_path):
    """-gff and -ai under torch.distributed.run: golden case 4 (two samples, one per rank) -- sample_miRge3.gff and the three
    a2IEditing files byte for byte what the reference wrote; rank 0 runs the two report kernels over the merged table."""
    case = GoldenCase("case4_gff_a2i")
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fake_bowtie")
    out = _run_cli_two_ranks(tmp_path, _case_fastq_files(case, tmp_path), case, ["-gff", "-ai", "-pbwt", fake, "-cpu", "1"], 29543)
    for f in ("miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv", "sample_miRge3.gff",
              "a2IEditing.report.csv", "a2IEditing.report.newform.csv", "a2IEditing.detail.txt"):
        assert (out / f).read_text() == case.text(f), f
    _same_report_but_total_input(out, case)


def test_cli_two_ranks_equal_one_process_at_scale(tmp_path, ci_libs):
    """The sharded run against the one-process run on the same two 1.5 M-read samples (synthetic libraries written as a
    miRge library directory): every output file byte for byte.  The one-process run formats mapped.csv / unmapped.csv on
    the GPU from the joint collapse of the raw reads; the sharded run merges the ranks' dictionaries with the weighted
    collapse and formats on the host -- two routes to the same sample matrix, the same row order, the same text."""
    from mirge3_amd.seqio import index_basename, write_fasta
    idx = tmp_path / "Libs" / ORG / "index.Libs"
    idx.mkdir(parents=True)
    (tmp_path / "Libs" / ORG / "annotation.Libs").mkdir()
    for key, lib in ci_libs.libs.items():
        write_fasta(str(idx / (index_basename(ORG, key, "miRBase") + ".fa")), lib)
    (tmp_path / "Libs" / ORG / "annotation.Libs" / f"{ORG}_merges_miRBase.csv").write_text("".join(",".join(r) + "\n" for r in ci_libs.merges))
    files = []
    for s in range(2):
        reads = synth.make_reads_chunked(ci_libs, 1_500_000, seed=900 + s)
        L = reads.lengths
        rec = FlatSeqs.join_columns([FlatSeqs.from_list(["@r"] * 1).take(np.zeros(len(reads), dtype=np.int64)), reads,
                                     FlatSeqs.from_list(["+"]).take(np.zeros(len(reads), dtype=np.int64)),
                                     FlatSeqs(np.full(int(L.sum()), ord("I"), dtype=np.uint8), reads.offsets)], b"\n\n\n\n")
        p = tmp_path / f"S{s + 1}.fastq"
        p.write_bytes(rec)
        files.append(str(p))
    case = SimpleNamespace(libdir=str(tmp_path / "Libs"))
    _run_cli(["-s", ",".join(files), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "one", "-shh", "-ie", "-tcf"])
    out2 = _run_cli_two_ranks(tmp_path, files, case, ["-ie", "-tcf"], 29547)
    names = ("annotation.report.csv", "annotation.report.html", "miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv", "isomirs.csv",
             "isomirs.samples.csv", "S1.trim.collapse.fa", "S2.trim.collapse.fa")
    first = (tmp_path / "one" / "S1.trim.collapse.fa").read_text().split("\n", 4)
    assert first[0].startswith(">seq1_") and first[2].startswith(">seq2_") and int(first[0][6:]) >= int(first[2][6:])
    for f in names:
        a, b = (tmp_path / "one" / f).read_bytes(), (out2 / f).read_bytes()
        assert a == b, f
    assert (tmp_path / "one" / "mapped.csv").stat().st_size > 20_000_000 and not (out2 / ".mirge_shards").exists()


def test_range_sample_and_split_of_a_dictionary(ctx, ci_libs):
    """mirge_reads_range_sample / mirge_reads_range_split (round 6: the sharded run's parallel tail): the quantile keys are
    keys of the dictionary in ascending order; the split orders the dictionary by (owner range, handle index), every read lands in
    the range its first 21 bases say, with its count, and the ranges are consecutive stretches of the sorted dictionary.  Reads with
    N, 32-50-nt reads, an empty range, one range, an empty dictionary."""
    from helpers import lex_key0
    from mirge3_amd import multigpu
    reads = synth.make_reads(ci_libs, 60000, seed=31, pool=9000, n_frac=0.03)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    u = raw.collapse()
    seqs = u.unpack().to_list()
    cnt, _ = u.counts()
    keys = np.array([lex_key0(q) for q in seqs], dtype=np.uint64)
    k = 64
    qs = u.range_sample(k)
    srt = np.sort(keys)
    assert np.array_equal(qs, srt[np.minimum(len(srt) - 1, (2 * np.arange(k) + 1) * len(srt) // (2 * k))])
    for n_parts in (1, 2, 5, 8):
        sp = multigpu.choose_splitters([(len(seqs), qs)], n_parts)
        assert sp.shape[0] == n_parts - 1 and (np.diff(sp.astype(np.int64)) >= 0).all()
        if n_parts == 5:
            sp[2] = sp[1]  # an empty range
        fs, c2, bounds = u.range_split(sp)
        got = fs.to_list()
        owner = np.searchsorted(sp, keys, side="right")
        assert bounds[0] == 0 and bounds[-1] == len(seqs)
        assert np.array_equal(np.diff(bounds), np.bincount(owner, minlength=n_parts))
        exp_order = np.argsort(owner, kind="stable")
        assert got == [seqs[i] for i in exp_order] and np.array_equal(c2, cnt[exp_order])
        if n_parts in (2, 8):  # balance: every range within a few 1/k of the dictionary of its share
            assert np.abs(np.diff(bounds) - len(seqs) / n_parts).max() <= 3 * len(seqs) / k + 50
        # consecutive stretches of the sorted dictionary: the largest sequence of a range sorts before the smallest of the next
        prev = None
        for q in range(n_parts):
            part = got[int(bounds[q]):int(bounds[q + 1])]
            if part:
                assert prev is None or prev < min(part)
                prev = max(part)
    u.close(); raw.close()
    empty = _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list([])).collapse()
    assert (empty.range_sample(8) == multigpu.KEY_NONE).all()
    fs, c2, bounds = empty.range_split(np.array([5, 9], dtype=np.uint64))
    assert len(fs) == 0 and list(bounds) == [0, 0, 0, 0]
    assert np.array_equal(multigpu.choose_splitters([(0, empty.range_sample(8))], 3), np.full(2, 1 << 63, dtype=np.uint64))
    empty.close()


def test_cli_ranks_write_their_ranges_of_the_per_read_tables(tmp_path):
    """The sharded CLI's parallel tail (fastpath.run_sharded_ranges): no rank builds the run's joint table alone -- every rank merges,
    annotates, orders and formats its RANGE of the sorted union and pwrites its stretch of mapped.csv / unmapped.csv.  Golden case 2
    (two samples, two ranks) and case 5 (three samples with the spike-in library on two ranks: rank 0 holds two samples): the
    reference's own files, byte for byte; the run log says which tail ran."""
    for name, port, extra in (("case2_two_samples", 29561, []), ("case5_three_samples_spikein", 29563, ["-spk"])):
        case = GoldenCase(name)
        d = tmp_path / name
        d.mkdir()
        out = _run_cli_two_ranks(d, _case_fastq_files(case, d), case, extra, port)
        for f in ("miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv"):
            assert (out / f).read_text() == case.text(f), (name, f)
        _same_report_but_total_input(out, case)
        log = (out / "run.log").read_text()
        assert '"tail": "ranges' in log and not (out / ".mirge_shards").exists()


def test_cli_ranges_tail_with_an_empty_range_and_an_idle_rank(tmp_path):
    """The parallel tail's corners: two samples that hold ONE sequence between them on three ranks -- one rank has no sample, the
    splitters coincide, two of the three key ranges are empty -- and a run whose samples share nothing.  The files are the
    one-process run's."""
    import subprocess
    import sys
    case = GoldenCase("case2_two_samples")
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    launcher = tmp_path / "run_cli.py"
    launcher.write_text("import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()\n" % root)
    one = case.seqs[0]
    sets = {"same": ([one] * 40, [one] * 25), "disjoint": (case.seqs[:30], case.seqs[30:70])}
    for tag, (a, b) in sets.items():
        d = tmp_path / tag
        d.mkdir()
        files = []
        for nm, seqs in (("S1", a), ("S2", b)):
            p = d / f"{nm}.fastq"
            p.write_text("".join(f"@r\n{q}\n+\n{'I' * len(q)}\n" for q in seqs))
            files.append(str(p))
        common = ["-s", ",".join(files), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(d), "-shh"]
        _run_cli(common + ["-dn", "one"])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
               "--master-port", "29569" if tag == "same" else "29573", str(launcher)] + common + ["-dn", "ranks"]
        r = subprocess.run(cmd, env=dict(os.environ, MIRGE_SHARE_GPU="1", OMP_NUM_THREADS="2"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        for f in ("annotation.report.csv", "miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv"):
            assert (d / "one" / f).read_bytes() == (d / "ranks" / f).read_bytes(), (tag, f)
        assert '"tail": "ranges' in (d / "ranks" / "run.log").read_text()


def test_cli_ranges_tail_equals_rank0_tail_and_one_process(tmp_path, ci_libs):
    """Three routes to the same files on three 0.7 M-read samples over three ranks that share the GPU: one process; the sharded run
    with rank 0 building the joint table alone (MIRGE_SHARD_TAIL=rank0, round 5); the sharded run with every rank writing its range
    (round 6).  Every output file byte for byte."""
    import subprocess
    import sys
    from mirge3_amd.seqio import index_basename, write_fasta
    idx = tmp_path / "Libs" / ORG / "index.Libs"
    idx.mkdir(parents=True)
    (tmp_path / "Libs" / ORG / "annotation.Libs").mkdir()
    for key, lib in ci_libs.libs.items():
        write_fasta(str(idx / (index_basename(ORG, key, "miRBase") + ".fa")), lib)
    (tmp_path / "Libs" / ORG / "annotation.Libs" / f"{ORG}_merges_miRBase.csv").write_text("".join(",".join(r) + "\n" for r in ci_libs.merges))
    files = []
    for s in range(3):
        reads = synth.make_reads_chunked(ci_libs, 700_000, seed=500 + s, n_frac=0.01)
        p = tmp_path / f"S{s + 1}.fastq"
        with open(p, "w") as fh:
            fh.write("".join(f"@r\n{q}\n+\n{'I' * len(q)}\n" for q in reads.to_list()))
        files.append(str(p))
    libdir = str(tmp_path / "Libs")
    _run_cli(["-s", ",".join(files), "-lib", libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "one", "-shh"])
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    launcher = tmp_path / "run_cli.py"
    launcher.write_text("import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()\n" % root)
    names = ("annotation.report.csv", "annotation.report.html", "miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv")
    for tail, port in (("rank0", 29565), ("ranges", 29567)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(launcher), "-s", ",".join(files), "-lib", libdir, "-on", ORG, "-db", "miRBase",
               "-o", str(tmp_path), "-dn", tail, "-shh"]
        r = subprocess.run(cmd, env=dict(os.environ, MIRGE_SHARE_GPU="1", OMP_NUM_THREADS="2", MIRGE_SHARD_TAIL=tail), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        for f in names:
            assert (tmp_path / "one" / f).read_bytes() == (tmp_path / tail / f).read_bytes(), (tail, f)
        log = (tmp_path / tail / "run.log").read_text()
        assert ('"tail": "ranges' in log) == (tail == "ranges")
    assert (tmp_path / "one" / "mapped.csv").stat().st_size > 10_000_000


def test_collapse_merge_equals_the_joint_collapse_of_the_raw_reads(ctx, ci_libs):
    """mirge_collapse_merge (round 6: what a run of several samples in one process calls): per-sample dictionaries merged on the device ==
    mirge_collapse with sample ids over the samples' concatenated raw reads (the route until round 5) == Python's Counters: the same
    unique reads, the same count matrix.  Samples of different kinds and sizes, one of them empty, reads with N and of 32-50 nt; and the
    CLI's files by either route (MIRGE_JOINT_COLLAPSE=raw) are the golden case's."""
    samples = [synth.make_reads(ci_libs, 150000, seed=41, pool=4000, n_frac=0.02), synth.make_reads_chunked(ci_libs, 90000, seed=42),
               FlatSeqs.from_list([]), synth.make_reads(ci_libs, 70000, seed=43, n_frac=0.05, long_frac=0.3)]
    raws = [_ffi.DeviceReads.pack(ctx, smp) for smp in samples]
    dicts = [r.collapse() for r in raws]
    merged = _ffi.DeviceReads.merge(ctx, dicts)
    allr = _ffi.DeviceReads.concat(ctx, raws)
    sid = np.repeat(np.arange(len(raws), dtype=np.int32), [len(r) for r in raws])
    joint = allr.collapse(sid, len(raws))
    assert len(merged) == len(joint) and merged.n_samples == joint.n_samples == 4

    def table(u):
        cnt, _ = u.counts()
        return dict(zip(u.unpack().to_list(), map(tuple, cnt.tolist())))
    tm, tj = table(merged), table(joint)
    assert tm == tj
    exp = [Counter(smp.to_list()) for smp in samples]
    assert set(tm) == set().union(*[set(c) for c in exp])
    for q, row in list(tm.items())[:3000]:
        assert list(row) == [c.get(q, 0) for c in exp]
    assert np.array_equal(merged.nonzero_per_sample(), [len(c) for c in exp])
    for h in dicts + raws + [merged, joint, allr]:
        h.close()


def test_weighted_collapse_merges_dictionaries(ctx, ci_libs):
    """mirge_collapse_weighted: three samples' dictionaries (unique reads + counts) merged == the joint collapse of their
    raw reads (counts matrix; first index = first entry of the concatenated dictionaries)."""
    from mirge3_amd.fastpath import merge_sample_reads
    from mirge3_amd.multigpu import SampleReads
    samples = [synth.make_reads(ci_libs, 20000, seed=70 + s, pool=3000, n_frac=0.02) for s in range(3)]
    parts = []
    for smp in samples:
        raw = _ffi.DeviceReads.pack(ctx, smp)
        u = raw.collapse()
        cnt, _ = u.counts()
        o = u.first_appearance_order()
        sq = u.unpack().take(o)
        parts.append(SampleReads.from_seqs(sq, cnt[o, 0]))
        assert parts[-1].lengths.dtype == np.uint8 and np.array_equal(parts[-1].offsets, sq.offsets)
        u.close(); raw.close()
    uniq = merge_sample_reads(ctx, parts)
    cnt, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    exp = [Counter(smp.to_list()) for smp in samples]
    assert set(seqs) == set().union(*[set(c) for c in exp]) and len(seqs) == len(set(seqs))
    for i, q in enumerate(seqs):
        assert [int(x) for x in cnt[i]] == [c.get(q, 0) for c in exp]
    allseq = [q for p in parts for q in FlatSeqs(p.data, p.offsets).to_list()]
    assert all(allseq[int(f)] == q for f, q in zip(first, seqs))
    assert all(allseq.index(q) == int(f) for f, q in list(zip(first, seqs))[:200])
    uniq.close()


def test_full_size_c2_route(ctx):
    """BASELINE configs[1] at its full size through the route `bench.py --workload c2` steps: 10 M raw reads, collapse and the
    exact mature-miRNA pass as ONE call (k_pass with the read count on the device), count join.  Conservation, the pass's
    subset rule (len < 26, 0 mismatches: manifoldAlign.py:85,93), agreement with the two-call sequence and with pass 0 of the
    full cascade read for read, and the oracle on a random sample."""
    sl = synth.make_libraries(seed=20260101, scale="full")
    c2 = Cascade(ctx, {"mirna": sl.libs["mirna"]}, n_pass=1)
    reads = synth.make_reads_chunked(sl, 10_000_000, seed=1000)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq, res = c2.collapse_and_run(raw)
    counts, first = uniq.counts()
    assert int(counts.sum()) == len(reads) and len(np.unique(first)) == len(uniq)
    ps, ref, off, mm = res.fetch()
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, 0, -2, len(sl.libs["mirna"]))
    c = counts[:, 0].astype(np.int64)
    assert cls[0, 0] == c[ps == 0].sum() and cls[0, 0] + c[ps < 0].sum() == len(reads) and ex.sum() == cls[0, 0] and iso.sum() == 0
    assert set(np.unique(ps).tolist()) <= {-1, 0} and (mm[ps == 0] == 0).all()
    useq = uniq.unpack()
    assert (useq.lengths[ps == 0] < 26).all()
    # the two-call sequence: same unique reads (as a set: the order of a collapse's output is unspecified), same answers
    u2 = raw.collapse(); r2 = c2.run(u2)
    cnt2, first2 = u2.counts()
    o1, o2 = np.argsort(first, kind="stable"), np.argsort(first2, kind="stable")
    assert np.array_equal(first[o1], first2[o2]) and np.array_equal(counts[o1], cnt2[o2])
    for x, y in zip((ps, ref, off, mm), r2.fetch()):
        assert np.array_equal(x[o1], y[o2])
    # pass 0 of the full cascade on a random sample of the collapsed reads, and the oracle on part of it
    rng = np.random.default_rng(4)
    pick = rng.permutation(len(uniq))[:300000]
    sub = useq.take(pick)
    casc = Cascade(ctx, sl.libs)
    g = casc.annotate(sub)
    assert np.array_equal(g[0] == 0, ps[pick] == 0)
    for x, y in zip(g[1:], (ref, off, mm)):
        assert np.array_equal(x[g[0] == 0], y[pick][g[0] == 0])
    k = 60000
    o = oracle.cascade(sub.data[: sub.offsets[k]], sub.offsets[: k + 1], oracle_libs_from(sl.libs)[:1], n_pass=1, indexed=True)
    for x, y in zip(o, (ps, ref, off, mm)):
        assert np.array_equal(x.astype(np.int64), y[pick][:k].astype(np.int64))
    for h in (res, r2, uniq, u2, raw, casc, c2):
        h.close()


def test_full_size_c3_properties(ctx):
    """BASELINE configs[2] at its full size -- 10 M raw reads against the human-sized libraries (130 Mb mRNA,
    11 Mb ncRNA): conservation, checksum of checksums, subset rules, order independence, and the oracle on
    a random sample of the collapsed reads."""
    sl = synth.make_libraries(seed=20260101, scale="full")
    casc = Cascade(ctx, sl.libs)
    reads = synth.make_reads_chunked(sl, 10_000_000, seed=1000)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq = raw.collapse()
    counts, first = uniq.counts()
    assert int(counts.sum()) == len(reads) and len(np.unique(first)) == len(uniq)
    res = casc.run(uniq)
    ps, ref, off, mm = res.fetch()
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, 0, 8, len(sl.libs["mirna"]))
    c = counts[:, 0].astype(np.int64)
    for p in range(9):
        assert cls[p, 0] == c[ps == p].sum()
    assert cls.sum() + c[ps < 0].sum() == len(reads)
    assert ex.sum() == cls[0, 0] and iso.sum() == cls[8, 0]
    useq = uniq.unpack()
    lens = useq.lengths
    assert (lens[ps == 0] < 26).all() and (lens[ps == 1] > 25).all()
    assert (mm[ps >= 0] >= 0).all() and (mm[ps >= 0] <= 2).all() and (mm[ps == 0] == 0).all() and (mm[ps == 3] == 0).all()
    rng = np.random.default_rng(2)
    pick = rng.permutation(len(uniq))[:300000]
    sub = useq.take(pick)
    g = casc.annotate(sub)
    for a, b in zip(g, (ps, ref, off, mm)):
        assert np.array_equal(a, b[pick])
    k = 40000
    o = oracle.cascade(sub.data[: sub.offsets[k]], sub.offsets[: k + 1], oracle_libs_from(sl.libs), n_pass=9, indexed=True)
    for a, b in zip(o, g):
        assert np.array_equal(a.astype(np.int64), b[:k].astype(np.int64))
    # row N2 at this size: every read of the exact-miRNA / isomiR passes (~0.7 M) typed by k_isotype; every record typed, the
    # exact rows that equal their canonical are ref_miRNA, and 15 000 records against the oracle's difflib restatement
    from mirge3_amd import gff
    mir, hp = sl.libs["mirna"], sl.libs["hairpin"]
    mseq, hseq = mir.seqs.to_list(), hp.seqs.to_list()
    pre_seq = dict(zip(hp.names, hseq))
    pre_of = {nm: hp.names[int(sl.mir_hairpin[q])] for q, nm in enumerate(mir.names)}
    tabs = gff.resolve_names(mir.names, dict(zip(mir.names, mseq)), pre_of, pre_seq)
    rows = np.nonzero((ps == 0) | (ps == 8))[0].astype(np.int64)
    recs = gff.isomir_records(casc, uniq, res, tabs, rows)
    assert len(recs) == len(rows) > 500000 and (recs["kind"] > 0).all()
    for q in rng.permutation(len(rows))[:15000]:
        i = int(rows[q])
        read, nm = useq.get(i), mir.names[ref[i]]
        kind, start, end, variant, cigar = oracle.gff_record(mseq[ref[i]], read, pre_seq[pre_of[nm]])
        r = recs[q]
        text = bytes(r["text"]).ljust(320, b"\0")
        got = ({1: "ref_miRNA", 2: "isomiR"}[int(r["kind"])], int(r["start"]), int(r["end"]), text[:r["vlen"]].decode(),
               text[r["vlen"]:r["vlen"] + r["clen"]].decode())
        assert got == (kind, start, end, variant, cigar), (read, mseq[ref[i]], got, (kind, start, end, variant, cigar))
    res.close(); uniq.close(); raw.close(); casc.close()


def test_full_size_c4_rank_shape(ctx):
    """BASELINE configs[3], what ONE rank of the 8-GPU run does: a 20 M-read sample (its own seed) through the
    one-call path (collapse with 16 384 partition buckets -> full cascade -> count join) against the human-sized
    libraries.  Conservation, checksum of checksums, subset rules, the one-call path == the two-call path, order
    independence (a sample of the unique reads annotated on their own) and the oracle on 40 k of them."""
    sl = synth.make_libraries(seed=20260101, scale="full")
    casc = Cascade(ctx, sl.libs)
    n = 20_000_000
    reads = synth.make_reads_chunked(sl, n, seed=1003)  # bench.py: seed = 1000 + rank
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq, res = casc.collapse_and_run(raw)
    counts, first = uniq.counts()
    assert int(counts.sum()) == n and len(np.unique(first)) == len(uniq) and first.max() < n
    ps, ref, off, mm = res.fetch()
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, 0, 8, len(sl.libs["mirna"]))
    c = counts[:, 0].astype(np.int64)
    for p in range(9):
        assert cls[p, 0] == c[ps == p].sum()
    assert cls.sum() + c[ps < 0].sum() == n
    assert ex.sum() == cls[0, 0] and iso.sum() == cls[8, 0]
    assert np.array_equal(ex[:, 0], np.bincount(ref[ps == 0], weights=c[ps == 0], minlength=len(ex)).astype(np.int64))
    useq = uniq.unpack()
    lens = useq.lengths
    assert (lens[ps == 0] < 26).all() and (lens[ps == 1] > 25).all()
    assert (mm[ps >= 0] >= 0).all() and (mm[ps >= 0] <= 2).all() and (mm[ps == 0] == 0).all() and (mm[ps == 3] == 0).all()
    # the reads themselves: every unique read is the raw read at its first index
    pick = np.random.default_rng(4).permutation(len(uniq))[:200000]
    sub = useq.take(pick)
    assert sub.to_list() == reads.take(first[pick]).to_list()
    # two calls give the same annotation, read for read (matched by first index: the unique order is unspecified)
    u2 = raw.collapse()
    r2 = casc.run(u2)
    c2, f2 = u2.counts()
    a2 = r2.fetch()
    o1, o2 = np.argsort(first), np.argsort(f2)
    assert np.array_equal(first[o1], f2[o2]) and np.array_equal(counts[o1], c2[o2])
    for x, y in zip((ps, ref, off, mm), a2):
        assert np.array_equal(x[o1], y[o2])
    r2.close(); u2.close()
    g = casc.annotate(sub)
    for a, b in zip(g, (ps, ref, off, mm)):
        assert np.array_equal(a, b[pick])
    k = 40000
    o = oracle.cascade(sub.data[: sub.offsets[k]], sub.offsets[: k + 1], oracle_libs_from(sl.libs), n_pass=9, indexed=True)
    for a, b in zip(o, g):
        assert np.array_equal(a.astype(np.int64), b[:k].astype(np.int64))
    res.close(); uniq.close(); raw.close(); casc.close()


def test_full_size_c5_properties(ctx):
    """BASELINE configs[4] at its full size: 50 M raw reads (32 768 partition buckets, the CS = 1024 LDS
    configuration of the collapse) -> exact + <=2-mismatch isomiR passes against the miRNA library -> count join ->
    per-position variant tally.  Conservation, subset rules, tally conservation (every accepted read is counted
    once per aligned position), and the oracle -- cascade AND variant tally -- on a 40 k sample of the
    unique reads."""
    from mirge3_amd import a2i
    sl = synth.make_libraries(seed=20260101, scale="full")
    libs = {"mirna": sl.libs["mirna"]}
    casc = Cascade(ctx, libs, n_pass=9)
    n = 50_000_000
    reads = synth.make_reads_chunked(sl, n, seed=1000)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq, res = casc.collapse_and_run(raw)
    counts, first = uniq.counts()
    assert int(counts.sum()) == n and len(np.unique(first)) == len(uniq)
    ps, ref, off, mm = res.fetch()
    assert set(np.unique(ps)) <= {-1, 0, 8}
    n_mirna = len(sl.libs["mirna"])
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, 0, 8, n_mirna)
    c = counts[:, 0].astype(np.int64)
    assert cls[0, 0] == c[ps == 0].sum() and cls[8, 0] == c[ps == 8].sum() and cls.sum() + c[ps < 0].sum() == n
    assert np.array_equal(iso[:, 0], np.bincount(ref[ps == 8], weights=c[ps == 8], minlength=n_mirna).astype(np.int64))
    t = a2i.tally(casc, uniq, res)
    assert (t["count_true"][:, 0] <= ex[:, 0] + iso[:, 0]).all() and (t["canon"] <= t["count_true"]).all()
    assert t["count_true"].sum() > n // 4 and (t["seq_true"] <= t["n_seqs"]).all()
    cen = t["census"]
    assert (cen[..., 2, :] <= cen[..., 1, :]).all() and (cen[..., 1, :] <= cen[..., 0, :]).all()
    assert cen[:, :, 0, 2, 2, 0].sum() > 0 and cen[:, 27:].sum() == 0
    useq = uniq.unpack()
    pick = np.random.default_rng(5).permutation(len(uniq))[:40000]
    sub = useq.take(pick)
    olibs = oracle_libs_from(sl.libs)
    o = oracle.cascade(sub.data, sub.offsets, [olibs[0]] + [None] * 7 + [olibs[8]], n_pass=9, indexed=True)
    for a, b in zip(o, (ps, ref, off, mm)):
        assert np.array_equal(a.astype(np.int64), b[pick].astype(np.int64))
    # the tally of the sample alone, GPU vs oracle restatement
    r_s = _ffi.DeviceReads.pack(ctx, sub)
    r_s.set_counts(counts[pick])
    res_s = casc.run(r_s)
    t_g = a2i.tally(casc, r_s, res_s, per_read=True)
    a_s = res_s.fetch()
    for x, y in zip(a_s, (ps, ref, off, mm)):
        assert np.array_equal(x, y[pick])
    mir = sl.libs["mirna"].seqs.to_list()
    t_o = oracle.variant_tally(sub.to_list(), counts[pick].astype(np.int64), a_s[0], a_s[1], np.arange(len(mir)), mir)
    for k in ("diag", "state", "n_seqs", "seq_true", "count_true", "canon", "kept_exact", "census"):
        assert np.array_equal(t_g[k], t_o[k]), k
    res_s.close(); r_s.close(); res.close(); uniq.close(); raw.close(); casc.close()


# ---------------------------------------------------------------- -ai / -gff against the reference's own files (case4)
GFF_A2I_CASES = ["case4_gff_a2i", "case6_gff_a2i"]  # (case 6: round 4, the same recipe on other libraries and reads)


def _case4_run(tmp_path, case_name="case4_gff_a2i", **flags):
    """a -gff / -ai golden case expanded to FASTQ files and pushed through the CLI's device-resident route"""
    from mirge3_amd import fastpath
    case = GoldenCase(case_name)
    files = []
    for s, nm in enumerate(case.samples):
        p = tmp_path / f"{nm}.fastq"
        with open(p, "w") as fh:
            for seq, row in zip(case.seqs, case.counts):
                if row[s]:
                    fh.write(f"@r\n{seq}\n+\n{'I' * len(seq)}\n" * int(row[s]))
        files.append(str(p))
    work = tmp_path / "out"
    work.mkdir()
    args = SimpleNamespace(libraries_path=case.libdir, organism_name=ORG, spikeIn=False, quiet=True, minimum_length=16,
                           crThreshold="0.1", device=0, isoform_entropy=False, threads=1, bowtieVersion="True", phred64=False,
                           bowtie_path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fake_bowtie"), **flags)
    out = fastpath.run(args, files, case.samples, str(work), "miRBase")
    return case, work, out


@pytest.mark.parametrize("case_name", GFF_A2I_CASES)
def test_a2i_report_equals_the_reference_files(tmp_path, case_name):
    """-ai: a2IEditing.report.csv, a2IEditing.report.newform.csv and a2IEditing.detail.txt byte for byte what the
    reference's a2i_editing wrote for the same reads (the two genome runs answered by the same bowtie stand-in), and
    the kernel's per-(miRNA, sample) counts against the oracle's string restatement."""
    case, work, out = _case4_run(tmp_path, case_name, AtoI=True)
    for f in ("miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv", "a2IEditing.report.csv",
              "a2IEditing.report.newform.csv", "a2IEditing.detail.txt"):
        assert (work / f).read_text() == case.text(f), f
    a = out["a2i"]
    with open(os.path.join(case.dir, "genome_retained.txt")) as fh:
        golden_ret = {ln.strip() for ln in fh if ln.strip() and not ln.startswith("#")}
    d = out["device"]
    seqs = d["seqs"].to_list()
    assert {s for s, r in zip(seqs, a["retained"]) if r} <= golden_ret
    # every gated (family, sample): counts, census and per-read states vs the restated reference functions
    t, fam = a["tally"], a["families"]
    from mirge3_amd.a2i import families, read_pseudo_fasta, BASE_PAIRS, CODE
    _, fam_of_ref = families(case.libs["mirna"].names, case.merges)
    pseudo = read_pseudo_fasta(os.path.join(case.libdir, ORG, "fasta.Libs", f"{ORG}_mirna_SNP_pseudo_miRBase.fa"))
    ps, ref = d["ann"][0], d["ann"][1]
    checked = 0
    for f, s in list(zip(*np.nonzero(a["gate"])))[:120]:  # (the string restatement is pure Python: the first 120 gated cells)
        rows = [i for i in d["order"] if ps[i] in (0, 8) and fam_of_ref[ref[i]] == f and t["state"][i] >= 0 and d["counts"][i, s] > 0]
        rows.sort(key=lambda i: ps[i] != 0)  # exact rows first, frame order inside
        reads = [seqs[i] for i in rows]
        cnts = [int(d["counts"][i, s]) for i in rows]
        o = oracle.a2i_group(pseudo[fam[f]], reads, cnts, {seqs[i] for i in rows if a["retained"][i]})
        assert o["diagonals"] == [int(t["diag"][i]) for i in rows] and o["states"] == [bool(t["state"][i]) for i in rows]
        assert (o["countSumTrue"], o["seqCountTrue"], o["canonicalSeqCount"]) == (t["count_true"][f, s], t["seq_true"][f, s], t["canon"][f, s])
        assert len(rows) == t["n_seqs"][f, s]
        for k, (x, y) in enumerate(BASE_PAIRS):
            assert o["census"][k] == t["census"][f, :, CODE[x], CODE[y], :, s].sum(axis=0).tolist()
        for q, c in o["count"].items():
            assert t["census"][f, q - 1, 0, 2, 2, s] == c
        checked += 1
    assert checked >= 20


def test_isotype_register_form_equals_the_array_form_on_the_device(tmp_path):
    """k_isotype types a read in registers (mirge_isotype_fast) and keeps the form on per-thread arrays for the pairs that one
    cannot take; MIRGE_ISO_FAST=0 sends every read through the arrays.  Both on the same 150 k-read sample (reads of both
    width classes, N calls), in fresh processes: the record arrays are equal byte for byte, and most reads are isomiRs."""
    import subprocess
    import sys
    ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import mirge3_amd\n"
        "from mirge3_amd import _ffi, synth, gff\n"
        "from mirge3_amd.cascade import Cascade\n"
        "sl = synth.make_libraries(seed=20260101, scale='small')\n"
        "ctx = _ffi.Context(0)\n"
        "casc = Cascade(ctx, sl.libs)\n"
        "reads = synth.make_reads(sl, 150000, seed=5, mix=dict(exact=0.25, isomir=0.55, hairpin=0.1, random=0.1), n_frac=0.02)\n"
        "uniq = _ffi.DeviceReads.pack(ctx, reads).collapse()\n"
        "res = casc.run(uniq)\n"
        "ps = res.fetch()[0]\n"
        "mir, hp = sl.libs['mirna'], sl.libs['hairpin']\n"
        "mseq, hseq = mir.seqs.to_list(), hp.seqs.to_list()\n"
        "pre_of = {nm: hp.names[int(sl.mir_hairpin[k])] for k, nm in enumerate(mir.names)}\n"
        "tabs = gff.resolve_names(mir.names, dict(zip(mir.names, mseq)), pre_of, dict(zip(hp.names, hseq)))\n"
        "rows = np.nonzero((ps == 0) | (ps == 8))[0].astype(np.int64)\n"
        "seqs = uniq.unpack().to_list()\n"
        "rows = np.array(sorted(rows.tolist(), key=lambda i: seqs[i]), dtype=np.int64)  # (the collapse's output order is not the same in two runs)\n"
        "recs = gff.isomir_records(casc, uniq, res, tabs, rows)\n"
        "np.save(sys.argv[1], recs.view(np.uint8).reshape(len(recs), -1))\n"
        "print('rows', len(rows), 'isomirs', int((recs['kind'] == 2).sum()))\n") % ROOT
    outs = []
    for fast in ("1", "0"):
        f = str(tmp_path / f"recs{fast}.npy")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, MIRGE_ISO_FAST=fast), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "rows" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
        outs.append((np.load(f), r.stdout))
    a, b = outs[0][0], outs[1][0]
    assert a.shape == b.shape and a.shape[0] > 3000 and a.shape[1] == 336
    bad = np.nonzero((a != b).any(axis=1))[0]
    assert bad.size == 0, (bad[:5], bytes(a[bad[0]]), bytes(b[bad[0]]))
    assert int(outs[0][1].split("isomirs")[1]) > a.shape[0] // 3


@pytest.mark.parametrize("case_name", GFF_A2I_CASES)
def test_gff_equals_the_reference_file(tmp_path, case_name):
    """-gff: sample_miRge3.gff byte for byte what the reference's create_gff wrote for the same reads (727 typed
    lines: every variant class, templated and non-templated additions, the three worked examples of
    summary.py:249-283, reads with N, names without annotation)."""
    case, work, out = _case4_run(tmp_path, case_name, gff_out=True)
    assert (work / "sample_miRge3.gff").read_text() == case.text("sample_miRge3.gff")
    # round 6: the default route formats the file on the device (mirge_gff_write_device): no record reaches the host
    assert out["gff"]["records"] is None and out["gff"]["lines"] == case.text("sample_miRge3.gff").count("\n") - 4
    # ... and round 5's route (records to the host, mirge_gff_write on its cores) writes the same file
    os.environ["MIRGE_GFF_DEVICE"] = "0"
    try:
        (tmp_path / "host").mkdir()
        case, work, out = _case4_run(tmp_path / "host", case_name, gff_out=True)
    finally:
        os.environ.pop("MIRGE_GFF_DEVICE")
    assert (work / "sample_miRge3.gff").read_text() == case.text("sample_miRge3.gff")
    recs = out["gff"]["records"]
    assert (recs["kind"] == 2).sum() > 500 and (recs["kind"] == 1).sum() > 40 and (recs["kind"] == 0).sum() > 0


# ---------------------------------------------------------------- read trimming (row N4)
def _trim_fastq(rng, n, adapter):
    """FASTQ records: inserts of 14-40 nt followed by the adapter in every state a sequencer produces (whole, cut off by
    the read length, with substitutions / a missing or extra base, absent, twice), qualities with bad tails, N ends"""
    recs = []
    for i in range(n):
        ins = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(14, 41))))
        kind = int(rng.integers(0, 10))
        ad = list(adapter)
        if kind in (1, 2):
            for _ in range(kind):
                ad[int(rng.integers(0, len(ad)))] = "ACGT"[int(rng.integers(0, 4))]
        elif kind == 3:
            del ad[int(rng.integers(1, len(ad) - 1))]
        elif kind == 4:
            ad.insert(int(rng.integers(1, len(ad) - 1)), "ACGT"[int(rng.integers(0, 4))])
        ad = "".join(ad)
        if kind == 5:
            seq = ins
        elif kind == 6:
            seq = ins + ad[:int(rng.integers(1, 12))]
        elif kind == 7:
            seq = ins + ad + "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(0, 9)))) + ad[:7]
        else:
            seq = ins + ad + "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(0, 6))))
        seq = seq[:int(rng.integers(36, 76))]
        if rng.random() < 0.05:
            seq = "N" * int(rng.integers(1, 3)) + seq[2:-1] + "N"
        q = np.full(len(seq), ord("I"), dtype=np.uint8)
        if rng.random() < 0.3:
            k = int(rng.integers(1, 15))
            q[-k:] = rng.integers(33, 50, size=min(k, len(seq)))
        if rng.random() < 0.1:
            q[:int(rng.integers(1, 5))] = 35
        if rng.random() < 0.1:
            q[int(rng.integers(0, len(seq)))] = 34
        recs.append((seq, q.tobytes().decode()))
    recs += [("ACGT", "IIII"), (adapter, "I" * len(adapter)), ("A" * 30, "#" * 30)]
    return recs


@pytest.mark.parametrize("opts", [
    dict(q_back=10),
    dict(q_back=10, adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAG"),
    dict(q_back=20, q_front=8, adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAG", trim_n=True, cut=[2, -1]),
    dict(q_back=10, adapter="AGATCGGAAGAGCNNNNACGT", error_rate=0.2, overlap=5, nextseq=20),
    dict(adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAG", cut=[-3]),
])
@pytest.mark.parametrize("per_modifier", [True, False])
def test_trimming_equals_the_restated_cutadapt_chain(ctx, opts, per_modifier):
    """mirge_reads_parse_trim (k_trim) against the oracle's full-matrix restatement of cutadapt's modifiers on the same
    FASTQ text: the collapsed dictionary -- sequences, counts, order of first appearance -- must be the worker's
    (digest.py:320-375), counted after every modifier (HEAD) or once.  Parity with cutadapt itself is unpinned."""
    rng = np.random.default_rng(len(str(opts)) + per_modifier)
    recs = _trim_fastq(rng, 6000, opts.get("adapter") or "TGGAATTCTCGGGTGCCAAGGAACTCCAG")
    text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
    trim = _ffi.MirgeTrim.make(adapter=opts.get("adapter"), quality_back=opts.get("q_back", -1), quality_front=opts.get("q_front", 0),
                               nextseq=opts.get("nextseq", -1), min_overlap=opts.get("overlap", 3), error_rate=opts.get("error_rate", 0.12),
                               trim_n=opts.get("trim_n", False), cut=opts.get("cut", []), count_per_modifier=per_modifier)
    raw, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, 16, trim)
    assert n_rec == len(recs)
    uniq = raw.collapse()
    cnt, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    order = np.argsort(first, kind="stable")
    got = [(seqs[i], int(cnt[i, 0])) for i in order]
    o = dict(opts)
    o.setdefault("q_back", None)
    want = oracle.trimmed_counts(recs, o, 16, per_modifier)
    assert got == list(want.items())
    assert len(raw) == sum(want.values()) and len(got) > 2000
    uniq.close(); raw.close()
    if opts.get("adapter") and "N" not in opts["adapter"]:  # the same chain on a FASTA text: no qualities, no quality modifiers
        ftext = "".join(f">r{i}\n{s}\n" for i, (s, q) in enumerate(recs)).encode()
        raw, n_rec = _ffi.DeviceReads.parse(ctx, ftext, 2, 16, trim)
        want = oracle.trimmed_counts([(s, None) for s, _ in recs], o, 16, per_modifier)
        uniq = raw.collapse()
        cnt, first = uniq.counts()
        seqs = uniq.unpack().to_list()
        order = np.argsort(first, kind="stable")
        assert [(seqs[i], int(cnt[i, 0])) for i in order] == list(want.items())
        uniq.close(); raw.close()


@pytest.mark.parametrize("opts", [dict(adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAG", times=3), dict(adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAG", indels=False),
                                  dict(adapter="GTTCAGAGTTCTACAGTCCGACGATC", front=True, indels=False, times=2),
                                  dict(adapters=[("back", "TGGAATTCTCGGGTGCCAAGGAACTCCAG"), ("front", "GTTCAGAGTTCTACAGTCCGACGATC")], times=2),
                                  dict(adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAG", read_wildcards=True),
                                  dict(adapter="TGGAATTCNNGGGTGCCAAGGAACTCCAG", adapter_wildcards=False),
                                  dict(adapter="TGGAATTCTCGGGTGCCAAGGAACTCCAG", action="none")])
def test_adapter_removal_repeated_and_without_indels(ctx, opts):
    """cutadapt's -n COUNT (AdapterCutter removes the best match, then searches what is left, up to COUNT times) and
    --no-indels (substitutions only): k_trim's general branch against the oracle's restatement, on the reads of the
    trimming test (adapters whole, partial, with substitutions / indels, twice in a read)."""
    rng = np.random.default_rng(len(str(opts)))
    a3, a5 = "TGGAATTCTCGGGTGCCAAGGAACTCCAG", "GTTCAGAGTTCTACAGTCCGACGATC"
    recs = _trim_fastq(rng, 3000, a3)
    recs += [(a5[k % 7:] + s[:40], q[:len(a5[k % 7:] + s[:40])].ljust(len(a5[k % 7:] + s[:40]), "I")) for k, (s, q) in enumerate(recs[:1500])]
    recs += [(s[:22] + a3[:12] + "AC" + a3, "I" * len(s[:22] + a3[:12] + "AC" + a3)) for s, _ in recs[:500]]  # the adapter twice
    recs += [(x, "I" * len(x)) for x in (s[:20] + a3[:5] + "N" + a3[6:9] + "NN" + a3[11:] for s, _ in recs[:500])]  # N in the adapter's copy
    text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
    ads = opts.get("adapters") or [("front" if opts.get("front") else "back", opts["adapter"])]
    a2 = ads[1] if len(ads) > 1 else (None, None)
    trim = _ffi.MirgeTrim.make(adapter=ads[0][1], front=ads[0][0] == "front", adapter2=a2[1], front2=a2[0] == "front", quality_back=10,
                               count_per_modifier=False, times=opts.get("times", 1), indels=opts.get("indels", True),
                               read_wildcards=opts.get("read_wildcards", False), adapter_wildcards=opts.get("adapter_wildcards", True),
                               action=opts.get("action", "trim"))
    raw, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, 16, trim)
    uniq = raw.collapse()
    cnt, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    order = np.argsort(first, kind="stable")
    want = oracle.trimmed_counts(recs, dict(q_back=10, **opts), 16, False)
    assert [(seqs[i], int(cnt[i, 0])) for i in order] == list(want.items()) and len(want) > 1000
    plain = oracle.trimmed_counts(recs, dict(q_back=10, **{k: v for k, v in opts.items() if k not in ("times", "indels", "read_wildcards", "adapter_wildcards", "action")}), 16, False)
    assert want != plain  # the option changes something on these reads
    uniq.close(); raw.close()


@pytest.mark.parametrize("kinds", [("back", "front"), ("front", "back"), ("back", "back"), ("front", "front")])
def test_two_adapters_best_match_equals_the_restated_adapter_cutter(ctx, kinds):
    """-a and -g in one run (mirge/__main__.py:65-74): cutadapt's AdapterCutter with times = 1 removes, per read, the ONE
    adapter that matches best (most matches, then fewest errors, then the first given).  k_trim's two-adapter branch
    against the oracle's restatement on reads that carry the 5' adapter, the 3' adapter, both, damaged copies, neither."""
    rng = np.random.default_rng(31 + len(kinds[0]) * 2 + len(kinds[1]))
    a3, a5 = "TGGAATTCTCGGGTGCCAAGGAACTCCAG", "GTTCAGAGTTCTACAGTCCGACGATC"
    ads = [(k, a5 if k == "front" else a3) for k in kinds]
    if kinds[0] == kinds[1]:
        ads[1] = (kinds[1], "ACGGTCAAGTCCATTGCA" if kinds[1] == "front" else "AGATCGGAAGAGCACACGTC")
    recs = []
    for i in range(4000):
        ins = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(15, 36))))
        parts = []
        for kind, ad in ads:
            x = list(ad)
            r = rng.random()
            if r < 0.15:
                x[int(rng.integers(0, len(x)))] = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.25:
                del x[int(rng.integers(1, len(x) - 1))]
            elif r < 0.45:
                x = x[int(rng.integers(3, len(x) - 3)):] if kind == "front" else x[:int(rng.integers(3, len(x) - 3))]
            elif r < 0.6:
                x = []
            parts.append("".join(x))
        seq = "".join(p for (k, _), p in zip(ads, parts) if k == "front") + ins + "".join(p for (k, _), p in zip(ads, parts) if k == "back")
        seq = seq[:int(rng.integers(40, 101))]
        recs.append((seq, "I" * len(seq)))
    text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
    trim = _ffi.MirgeTrim.make(adapter=ads[0][1], front=ads[0][0] == "front", adapter2=ads[1][1], front2=ads[1][0] == "front",
                               quality_back=10, count_per_modifier=False)
    raw, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, 16, trim)
    uniq = raw.collapse()
    cnt, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    order = np.argsort(first, kind="stable")
    want = oracle.trimmed_counts(recs, dict(q_back=10, adapters=ads), 16, False)
    assert [(seqs[i], int(cnt[i, 0])) for i in order] == list(want.items()) and len(want) > 1000
    # the choice is exercised: some reads lose the first adapter, some the second, and both differ from either alone
    alone = [oracle.trimmed_counts(recs, dict(q_back=10, adapters=[a]), 16, False) for a in ads]
    assert want != alone[0] and want != alone[1]
    uniq.close(); raw.close()


@pytest.mark.parametrize("form", ["-g A...B", "-a A...B", "-g ^A...B$", "-g ^A", "-a B$", "-g ^A -a B$", "-a A...B -n 2 --no-indels"])
def test_linked_and_anchored_adapters_equal_the_restated_cutadapt(ctx, form):
    """Round 5: the specification strings the reference hands to cutadapt unchanged (digest.py:66-84) -- a linked adapter
    `A...B` (docs/source/quick_start.md:213-220 shows one under -g), anchored `^A` / `B$` -- through k_trim's general branch
    against the oracle's restatement of LinkedAdapter.match_to / PrefixAdapter / SuffixAdapter, on reads that carry the 5' part
    at the first base, inside, damaged, partial or not at all, and the same for the 3' part.  Parity with cutadapt: unpinned."""
    from mirge3_amd.cli import parse_args
    from mirge3_amd.collapse import trim_from_args, parse_adapter_spec, adapters_from_args
    rng = np.random.default_rng(sum(map(ord, form)))
    a5, a3 = "TTAGGCACGT", "TGGAATTCTCGGGTGCCAAGGAACTCCAGT"
    recs = []
    for i in range(5000):
        ins = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(15, 36))))
        x5, x3 = list(a5), list(a3)
        r = rng.random()
        if r < 0.15:
            x5[int(rng.integers(0, len(x5)))] = "ACGT"[int(rng.integers(0, 4))]
        elif r < 0.25:
            del x5[int(rng.integers(1, len(x5) - 1))]
        elif r < 0.35:
            x5 = x5[int(rng.integers(1, len(x5) - 2)):]
        elif r < 0.5:
            x5 = []
        elif r < 0.6:
            x5 = list("ACGT"[int(rng.integers(0, 4))] * int(rng.integers(1, 4))) + x5  # the 5' part not at the first base
        r = rng.random()
        if r < 0.15:
            x3[int(rng.integers(0, len(x3)))] = "ACGT"[int(rng.integers(0, 4))]
        elif r < 0.25:
            del x3[int(rng.integers(1, len(x3) - 1))]
        elif r < 0.45:
            x3 = x3[:int(rng.integers(3, len(x3) - 3))]
        elif r < 0.6:
            x3 = []
        elif r < 0.7:
            x3 = x3 + list("ACGT"[int(rng.integers(0, 4))] * int(rng.integers(1, 5)))   # something behind the 3' part
        seq = "".join(x5) + ins + "".join(x3)
        recs.append((seq, "I" * len(seq)))
    recs += [("", ""), (a5, "I" * len(a5)), (a3, "I" * len(a3)), (a5 + a3, "I" * (len(a5) + len(a3)))]
    text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
    argv = ["-s", "x.fq", "-lib", "/x", "-on", "human", "--trim-count", "once"]
    for tok in form.replace("A...B", f"{a5}...{a3}").replace("^A", "^" + a5).replace("B$", a3 + "$").split():
        argv.append(tok)
    args = parse_args(argv)
    trim = trim_from_args(args)
    specs = [parse_adapter_spec(k, q) for k, q in adapters_from_args(args)]
    o = dict(q_back=10, times=int(args.times or 1), indels=bool(args.indels))
    if specs[0].get("linked"):
        o["linked"] = specs[0]
    else:
        o["adapters"] = [(sp["kind"], sp["seq"], sp["anchored"]) for sp in specs]
    raw, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, 16, trim)
    assert n_rec == len(recs)
    uniq = raw.collapse()
    cnt, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    order = np.argsort(first, kind="stable")
    want = oracle.trimmed_counts(recs, o, 16, False)
    got = [(seqs[i], int(cnt[i, 0])) for i in order]
    assert got == list(want.items()) and len(want) > 1500
    # the form matters: the regular two-adapter chain on the same reads counts differently
    plain = oracle.trimmed_counts(recs, dict(q_back=10, adapters=[("front", a5), ("back", a3)]), 16, False)
    assert want != plain
    uniq.close(); raw.close()


@pytest.mark.parametrize("per_modifier", [True, False])
def test_five_prime_adapter_equals_the_restated_search(ctx, per_modifier):
    """-g: k_trim<64, false, true> against oracle.adapter_locate_front inside the modifier chain -- the adapter whole,
    with its first bases missing, with a substitution / a missing / an extra base, behind a few other bases, absent, twice,
    and at the read's end; adapters of several lengths; with quality trimming in front and N trimming + a cut behind."""
    rng = np.random.default_rng(11 + per_modifier)
    full = "GTTCAGAGTTCTACAGTCCGACGATCTGGAATTCTCGGGTGCCAAGGAACTCCAGTCACACGTC"
    for m in (5, 12, 26, 33, 64):
        ad = full[:m]
        recs = []
        for i in range(1500):
            ins = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(14, 41))))
            x = list(ad)
            kind = i % 9
            if kind == 1:
                x = x[int(rng.integers(1, max(2, m - 2))):]
            elif kind == 2 and m > 4:
                x[int(rng.integers(0, m))] = "ACGT"[int(rng.integers(0, 4))]
            elif kind == 3 and m > 6:
                del x[int(rng.integers(1, m - 1))]
            elif kind == 4 and m > 6:
                x.insert(int(rng.integers(1, m - 1)), "ACGT"[int(rng.integers(0, 4))])
            x = "".join(x)
            if kind == 5:
                seq = ins
            elif kind == 6:
                seq = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(1, 7)))) + x + ins
            elif kind == 7:
                seq = x + ins[:10] + x + ins
            elif kind == 8:
                seq = ins + x
            else:
                seq = x + ins
            if rng.random() < 0.05:
                seq = seq + "NN"
            q = np.full(len(seq), ord("I"), dtype=np.uint8)
            if rng.random() < 0.2:
                q[-int(rng.integers(1, 6)):] = 34
            recs.append((seq, q.tobytes().decode()))
        text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
        trim = _ffi.MirgeTrim.make(adapter=ad, front=True, quality_back=10, trim_n=True, cut=[-1], count_per_modifier=per_modifier)
        raw, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, 12, trim)
        assert n_rec == len(recs)
        uniq = raw.collapse()
        cnt, first = uniq.counts()
        seqs = uniq.unpack().to_list()
        order = np.argsort(first, kind="stable")
        got = [(seqs[i], int(cnt[i, 0])) for i in order]
        want = oracle.trimmed_counts(recs, dict(q_back=10, adapter=ad, front=True, trim_n=True, cut=[-1]), 12, per_modifier)
        assert got == list(want.items()), m
        assert len(got) > 800
        uniq.close(); raw.close()
    with pytest.raises(RuntimeError):
        _ffi.DeviceReads.parse(ctx, b"@r\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n", 1, 12, _ffi.MirgeTrim.make(adapter="ACGNT", front=True))


def test_trimming_300_cycle_lines_and_every_adapter_length(ctx):
    """Lines longer than 255 characters (the origin column of the DP entry is 15 bits wide, the packed read still <= 255 nt
    after trimming) and one adapter of every length 1-64 (k_trim<M, true> is a kernel per length; with an N the general
    k_trim<64, false>) against the oracle's full-matrix restatement."""
    rng = np.random.default_rng(300)
    full = "TGGAATTCTCGGGTGCCAAGGAACTCCAGTCACACGTCTGAACTCCAGTCACGATCGGAAGAGCAC"[:64]
    def collapsed(recs, trim):
        text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
        raw, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, 10, trim)
        assert n_rec == len(recs)
        uniq = raw.collapse()
        cnt, first = uniq.counts()
        seqs = uniq.unpack().to_list()
        order = np.argsort(first, kind="stable")
        got = [(seqs[i], int(cnt[i, 0])) for i in order]
        uniq.close(); raw.close()
        return got
    # 300-cycle lines: insert + adapter + read-through (a chance match of the adapter's first bases inside the tail is
    # a legitimate earlier/later candidate for both sides), and lines without the adapter at all
    ad = full[:29]
    recs = []
    for i in range(1500):
        L = int(rng.choice([18, 22, 30, 120, 250, 254]))
        ins = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=L))
        tail = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=300))
        seq = (ins + ad + tail)[:300] if i % 7 else ins
        recs.append((seq, "I" * len(seq)))
    opts = dict(adapter=ad, q_back=10)
    want = oracle.trimmed_counts(recs, opts, 10, False)
    got = collapsed(recs, _ffi.MirgeTrim.make(adapter=ad, quality_back=10, count_per_modifier=False))
    assert got == list(want.items()) and len(got) > 500
    for m in list(range(1, 65)) + ["N"]:
        a = full[:m] if m != "N" else full[:12] + "NN" + full[14:40]
        recs = []
        for i in range(300):
            ins = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(15, 36))))
            x = list(a)
            if i % 3 == 1 and len(x) > 4:
                x[int(rng.integers(0, len(x)))] = "ACGT"[int(rng.integers(0, 4))]
            if i % 5 == 2 and len(x) > 6:
                del x[int(rng.integers(1, len(x) - 1))]
            seq = (ins + "".join(x) + "ACGTTGCA")[:int(rng.integers(40, 110))]
            recs.append((seq, "I" * len(seq)))
        want = oracle.trimmed_counts(recs, dict(adapter=a, q_back=10), 10, False)
        got = collapsed(recs, _ffi.MirgeTrim.make(adapter=a, quality_back=10, count_per_modifier=False))
        assert got == list(want.items()), m


def test_trimmed_and_umi_sliced_reads_beyond_255_nt(ctx, tmp_path):
    """What is left of a line after trimming -- or after the UMIs are cut off -- may be longer than 255 nt (a merged pair, a 2 x 300
    run with its adapter at the very end): those reads go to the long read class instead of stopping the run.
    Lines of 270-700 characters: insert of 200-600 nt, adapter (exact, one substitution, one deletion), read-through or none, low
    quality at either end, duplicates; k_trim and the UMI slicing against the oracle's restatements."""
    rng = np.random.default_rng(255)
    ad = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"
    inserts = ["".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(L))) for L in rng.choice([200, 250, 255, 256, 257, 300, 420, 600], size=90)]
    recs = []
    for i in range(1400):
        ins = inserts[int(rng.integers(0, len(inserts)))]
        a = list(ad)
        kind = int(rng.integers(0, 7))
        if kind == 1:
            a[int(rng.integers(0, len(a)))] = "ACGT"[int(rng.integers(0, 4))]
        elif kind == 2:
            del a[int(rng.integers(1, len(a) - 1))]
        tail = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(0, 80))))
        seq = ins if kind == 3 else (ins + "".join(a)[:int(rng.integers(2, 12))] if kind == 4 else ins + "".join(a) + tail)
        q = np.full(len(seq), ord("I"), dtype=np.uint8)
        if rng.random() < 0.3:
            k = int(rng.integers(1, 40))
            q[-k:] = rng.integers(33, 48, size=k)
        if rng.random() < 0.1:
            q[:int(rng.integers(1, 6))] = 35
        recs.append((seq, q.tobytes().decode()))
    text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
    for opts in (dict(adapter=ad, q_back=10), dict(adapter=ad, q_back=20, q_front=8, cut=[3, -2]), dict(q_back=15)):
        for per_modifier in (False, True):
            trim = _ffi.MirgeTrim.make(adapter=opts.get("adapter"), quality_back=opts["q_back"], quality_front=opts.get("q_front", 0),
                                       cut=opts.get("cut", []), count_per_modifier=per_modifier)
            raw, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, 16, trim)
            assert n_rec == len(recs)
            gc = raw.group_counts()
            uniq = raw.collapse()
            cnt, first = uniq.counts()
            seqs = uniq.unpack().to_list()
            order = np.argsort(first, kind="stable")
            got = [(seqs[i], int(cnt[i, 0])) for i in order]
            want = oracle.trimmed_counts(recs, dict(opts), 16, per_modifier)
            assert got == list(want.items())
            n_long = sum(c for q_, c in want.items() if len(q_) > 255)
            assert n_long > 300 and int(gc[4] + gc[9]) == n_long and sum(c for q_, c in want.items() if len(q_) <= 255) > 100
            uniq.close(); raw.close()
    # -umi 5,3 [-udd] on the same kind of lines: [5 nt] insert [3 nt] adapter
    umis = ["".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=8)) for _ in range(7)]
    urecs = []
    for i in range(1200):
        ins = inserts[int(rng.integers(0, 40))]
        u = umis[int(rng.integers(0, len(umis)))]
        seq = u[:5] + ins + u[5:] + ad + "ACGTAC"[:int(rng.integers(0, 7))]
        q = np.full(len(seq), ord("I"), dtype=np.uint8)
        if rng.random() < 0.2:
            q[-int(rng.integers(1, 10)):] = 34
        urecs.append((seq, q.tobytes().decode()))
    utext = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(urecs)).encode()
    for dedup in (False, True):
        trim = _ffi.MirgeTrim.make(adapter=ad, quality_back=10, count_per_modifier=False)
        got, n, n_rec, csv = _umi_dict(ctx, utext, 1, 16, trim, _ffi.MirgeUmi.make(5, 3, dedup=dedup), tmp_path, f"long{int(dedup)}")
        keys = oracle.umi_worker_reads(urecs, dict(q_back=10, q_front=0, adapter=ad), 5, 3, 16, False, False)
        want, trimmed, rows = oracle.umi_baking(keys, 5, 3, 16, dedup)
        assert n_rec == len(urecs) and n == trimmed and got == want
        assert sum(len(k) > 255 for k, _ in want) > 10 and csv == ("".join(rows) if dedup else None)


def test_handles_outliving_their_context_do_not_crash(tmp_path):
    """A read set, a result or a library still referenced when its context is closed (a traceback holds it, the collector comes
    late) is closed with the context; its own close / finalizer afterwards is a no-op.  Found as a segmentation fault at the end of
    a pytest run whose failing test had left two read sets open.  In a child process: a regression would take the suite down."""
    import subprocess
    import sys
    ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import mirge3_amd as m\n"
        "from mirge3_amd import _ffi, synth\n"
        "from mirge3_amd.cascade import Cascade\n"
        "from mirge3_amd.seqio import FlatSeqs\n"
        "ctx = _ffi.Context(0)\n"
        "sl = synth.make_libraries(seed=3, scale='ci')\n"
        "casc = Cascade(ctx, sl.libs)\n"
        "raw = _ffi.DeviceReads.pack(ctx, synth.make_reads(sl, 5000, seed=1))\n"
        "uniq = raw.collapse()\n"
        "res = casc.run(uniq)\n"
        "n = len(uniq)\n"
        "ctx.close()\n"
        "for o in (res, uniq, raw): o.close()\n"
        "del res, uniq, raw, casc\n"
        "import gc; gc.collect()\n"
        "ctx2 = _ffi.Context(0)\n"
        "r2 = _ffi.DeviceReads.pack(ctx2, FlatSeqs.from_list(['ACGTACGTACGTACGTACGT'] * 3))\n"
        "assert len(r2.collapse()) == 1\n"
        "print('alive', n)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "alive" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]


def test_cli_with_adapter_trimming_end_to_end(tmp_path):
    """`-a illumina` end to end: golden case 1's reads with the adapter appended and cut at 50 nt come out as the
    reference's tables when the fully trimmed read is counted once (--trim-count once)."""
    case = GoldenCase("case1_single")
    ad = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"
    p = tmp_path / "S1.fastq"
    with open(p, "w") as fh:
        for seq, row in zip(case.seqs, case.counts):
            # sequences in which cutadapt would find the adapter's first bases on their own are left out of this check
            if oracle.adapter_locate_back(ad, seq) is not None:
                continue
            s = (seq + ad)[:50]
            if oracle.trim_stages(s, "I" * len(s), dict(q_back=10, adapter=ad))[-1] != seq:
                continue
            fh.write(f"@r\n{s}\n+\n{'I' * len(s)}\n" * int(row[0]))
    _run_cli(["-s", str(p), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "trimmed", "-shh", "-a", "illumina",
              "--trim-count", "once"])
    # the same reads already trimmed, no adapter option
    p2 = tmp_path / "S1b.fastq"
    with open(p2, "w") as fh:
        for ln in open(tmp_path / "trimmed" / "mapped.csv").read().splitlines()[1:] + open(tmp_path / "trimmed" / "unmapped.csv").read().splitlines()[1:]:
            f = ln.split(",")
            fh.write(f"@r\n{f[0]}\n+\n{'I' * len(f[0])}\n" * int(f[-1]))
    (tmp_path / "S1b.fastq").rename(tmp_path / "plain" / "S1.fastq") if (tmp_path / "plain").mkdir() is None else None
    _run_cli(["-s", str(tmp_path / "plain" / "S1.fastq"), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "untrimmed", "-shh"])
    for f in ("miR.Counts.csv", "miR.RPM.csv"):
        assert (tmp_path / "trimmed" / f).read_text() == (tmp_path / "untrimmed" / f).read_text(), f
    a = (tmp_path / "trimmed" / "annotation.report.csv").read_text().splitlines()[1].split(",")
    b = (tmp_path / "untrimmed" / "annotation.report.csv").read_text().splitlines()[1].split(",")
    assert a[1:] == b[1:] and int(a[2]) > 1000


def test_cli_documented_linked_adapter_command_line(tmp_path):
    """docs/source/quick_start.md:213-220, "Trimming both 5' and 3' adapters - Linked adapters": the reference's own example
    `-g "TTAGGC...TGGAATTCTCGGGTGCCAAGGAACTCCAGT"` (refused until round 5: the dots reached the C ABI as adapter letters).
    Golden case 1's reads between the two adapters, some with a damaged or missing part; the run must equal the run of the
    reads the oracle's restatement of cutadapt's LinkedAdapter leaves (`-g`: both parts required, else the read stays), and
    the run log must name the option as one not yet compared with cutadapt itself."""
    case = GoldenCase("case1_single")
    a5, a3 = "TTAGGC", "TGGAATTCTCGGGTGCCAAGGAACTCCAGT"
    lk = dict(front=a5, back=a3, front_anchored=False, back_anchored=False, front_required=True, back_required=True)
    rng = np.random.default_rng(4)
    recs = []
    for seq, row in zip(case.seqs, case.counts):
        for _ in range(int(row[0])):
            r = rng.random()
            s = a5 + seq + a3[:int(rng.integers(8, len(a3) + 1))]
            if r < 0.05:
                s = seq + a3          # no 5' part: required, so the read stays whole
            elif r < 0.10:
                s = a5 + seq          # no 3' part: required as well under -g
            elif r < 0.15:
                s = "AC" + a5 + seq + a3  # the regular 5' part is found inside the read too
            recs.append(s[:75])
    p = tmp_path / "S1.fastq"
    p.write_text("".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(recs)))
    _run_cli(["-s", str(p), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "linked", "-g", f"{a5}...{a3}",
              "--trim-count", "once"])
    want = [oracle.trim_stages(s, "I" * len(s), dict(q_back=10, linked=lk))[-1] for s in recs]
    assert sum(w != s for w, s in zip(want, recs)) > 0.8 * len(recs) and sum(w == s for w, s in zip(want, recs)) > 0.05 * len(recs)
    (tmp_path / "plain").mkdir()
    p2 = tmp_path / "plain" / "S1.fastq"
    p2.write_text("".join(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(want)))
    _run_cli(["-s", str(p2), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "plain_out", "-shh"])
    for f in ("miR.Counts.csv", "miR.RPM.csv", "mapped.csv", "unmapped.csv", "annotation.report.csv"):
        assert (tmp_path / "linked" / f).read_text() == (tmp_path / "plain_out" / f).read_text(), f
    log = (tmp_path / "linked" / "run.log").read_text()
    assert "a linked adapter (A...B)" in log and "have not been compared with cutadapt itself" in log


def test_reads_longer_than_255_nt(ctx, ci_libs, ci_cascade, tmp_path):
    """The reference puts no upper bound on a read (`-M` is parsed and never read, parse.py:102; the worker tests the minimum
    only, digest.py:348,368): an untrimmed 300-cycle read, or a merged pair, goes through bowtie like any other.  Reads of
    256 / 300 / 600 / 5000 nt -- fragments of mRNA, rRNA and ncRNA references with 0-3 mismatches placed inside and outside
    the seed, with an N, with a T tail, a read that runs over a reference's end, random ones -- mixed into ordinary reads:
    pack and parse, unpack, collapse (duplicates, several samples, weights), the cascade against the oracle (with and without
    the mRNA library), the count join, the sorted order, both CSV routes and the CLI."""
    rng = np.random.default_rng(77)
    libs = ci_libs.libs

    def mutate(q, where):
        q = list(q)
        for w in where:
            q[w] = "ACGT"[("ACGT".index(q[w]) + 1 + int(rng.integers(0, 3))) % 4] if q[w] in "ACGT" else "A"
        return "".join(q)

    longs = []
    for key, n_take in (("mrna", 14), ("rrna", 4), ("ncrna_others", 6)):
        seqs_ = [q for q in libs[key].seqs.to_list() if len(q) >= 700]
        for k in range(min(n_take, len(seqs_))):
            ref = seqs_[k]
            for L in (256, 300, 600):
                o = int(rng.integers(0, len(ref) - L))
                frag = ref[o:o + L]
                longs += [frag, mutate(frag, [5]), mutate(frag, [40, 200]), mutate(frag, [3, 90]), mutate(frag, [100, 150, 250]),
                          mutate(frag, [27]), mutate(frag, [28]), frag[:120] + "N" + frag[121:], frag[:10] + "N" + frag[11:]]
            longs.append(ref[-300:] + "ACGTACGTAC")       # runs over the reference's end
            longs.append(ref[:280] + "T" * 12)            # a T tail: pass 3 strips it, the others see it
    big = [q for q in libs["mrna"].seqs.to_list() if len(q) >= 5200]
    if big:
        longs += [big[0][100:5100], mutate(big[0][100:5100], [4000, 4500])]
    longs += ["".join("ACGT"[x] for x in rng.integers(0, 4, 400)) for _ in range(5)] + ["T" * 300, "A" * 256 + "TTTT", "ACGT" * 64]
    longs = [q for q in longs if set(q) <= set("ACGTN")]
    shorts = synth.make_reads(ci_libs, 3000, seed=5, n_frac=0.02).to_list()
    mixed = shorts[:1500] + longs + shorts[1500:] + longs[:25] + [longs[3]] * 4   # duplicates of long reads, far apart
    order = rng.permutation(len(mixed))
    mixed = [mixed[i] for i in order]
    fs = FlatSeqs.from_list(mixed)
    # pack (host ASCII) and parse (text on the device) agree, unpack returns the letters
    dr = _ffi.DeviceReads.pack(ctx, fs)
    text = "".join(f"@r{i}\n{q}\n+\n{'I' * len(q)}\n" for i, q in enumerate(mixed)).encode()
    dp, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, 16)
    assert n_rec == len(mixed) and len(dr) == len(dp) == len(mixed)
    assert dr.unpack().to_list() == mixed and dp.unpack().to_list() == mixed
    gc = dr.group_counts()
    assert len(gc) == 10 and int(gc[4] + gc[9]) == sum(len(q) > 255 for q in mixed) and np.array_equal(gc, dp.group_counts())
    # collapse: the dictionary of the reference (digest.py:141-163)
    for raw in (dr, dp):
        u = raw.collapse()
        cnt, first = u.counts()
        o_first, o_cnt, _ = oracle.collapse(fs.data, fs.offsets)
        od = np.argsort(first, kind="stable")
        assert len(u) == len(o_first) and np.array_equal(first[od], o_first) and np.array_equal(cnt[od, 0].astype(np.int64), o_cnt)
        useq = u.unpack()
        assert [useq.to_list()[i] for i in od] == [mixed[i] for i in o_first]
        # the cascade, every field, against the oracle -- all nine passes, then without the mRNA library (pass 7 absent)
        res = ci_cascade.run(u)
        got = res.fetch()
        want = oracle.cascade(useq.data, useq.offsets, oracle_libs_from(libs), n_pass=9, indexed=False)
        for a, b in zip(want, got):
            assert np.array_equal(a.astype(np.int64), b.astype(np.int64))
        lens = useq.lengths
        assert (got[0][lens > 255] >= 0).sum() > 50 and (got[0][lens > 255] < 0).sum() >= 5   # both kinds are present
        assert set(np.unique(got[0][lens > 255]).tolist()) & {4, 5, 6, 7}
        cls, ex, iso = _ffi.count_join(ctx, u, res, 0, 8, len(libs["mirna"]))
        for p_ in range(9):
            assert cls[p_, 0] == cnt[got[0] == p_, 0].sum()
        # the rows pandas would sort (several samples) and both CSV routes
        so = u.sorted_order()
        ul = useq.to_list()
        assert [ul[i] for i in so] == sorted(ul)
        res.close(); u.close()
    no_mrna = {k: v for k, v in libs.items() if k != "mrna"}
    c2 = Cascade(ctx, no_mrna)
    ul = FlatSeqs.from_list(sorted(set(longs)))
    got = c2.annotate(ul)
    olibs = oracle_libs_from(libs)
    olibs[7] = (np.zeros(0, np.uint8), np.zeros(1, np.int64))
    want = oracle.cascade(ul.data, ul.offsets, olibs, n_pass=9, indexed=False)
    for a, b in zip(want, got):
        assert np.array_equal(a.astype(np.int64), b.astype(np.int64))
    assert (got[0] == 7).sum() == 0
    c2.close()
    # several samples + weights (the sharded run's merge)
    half = len(mixed) // 2
    u2 = collapse_samples(ctx, [FlatSeqs.from_list(mixed[:half]), FlatSeqs.from_list(mixed[half:])])
    c2_, _ = u2.counts()
    sq = u2.unpack().to_list()
    ca, cb = Counter(mixed[:half]), Counter(mixed[half:])
    assert len(sq) == len(set(mixed)) and all((int(c2_[i, 0]), int(c2_[i, 1])) == (ca.get(q, 0), cb.get(q, 0)) for i, q in enumerate(sq))
    u2.close(); dr.close(); dp.close()
    # the CLI on a file that holds them: every row of mapped.csv / unmapped.csv as the DataFrame route writes it
    case = GoldenCase("case1_single")
    cl = [q for q in case.libs["mrna"].seqs.to_list() if len(q) >= 320][:3]
    extra = [cl[0][5:305], cl[1][0:256], mutate(cl[2][10:310], [100]), "".join("ACGT"[x] for x in rng.integers(0, 4, 333))] if len(cl) == 3 else \
        ["".join("ACGT"[x] for x in rng.integers(0, 4, 333)), "".join("ACGT"[x] for x in rng.integers(0, 4, 256))]
    files = _case_fastqs(case, tmp_path)
    with open(files[0], "a") as fh:
        for k, q in enumerate(extra * 2):
            fh.write(f"@long{k}\n{q}\n+\n{'I' * len(q)}\n")
    _run_cli(["-s", files[0], "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "withlong", "-shh"])
    rows = {}
    for f in ("mapped.csv", "unmapped.csv"):
        for ln in (tmp_path / "withlong" / f).read_text().splitlines()[1:]:
            rows[ln.split(",")[0]] = (f, ln)
    ol = oracle.cascade(FlatSeqs.from_list(extra).data, FlatSeqs.from_list(extra).offsets, oracle_libs_from(case.libs), n_pass=9, indexed=False)
    for q, p_ in zip(extra, ol[0]):
        f, ln = rows[q]
        assert f == ("mapped.csv" if p_ >= 0 else "unmapped.csv") and ln.endswith(",2") and ln.split(",")[1] == ("1" if p_ >= 0 else "0")
    golden = {ln.split(",")[0]: ln for f in ("mapped.csv", "unmapped.csv") for ln in case.text(f).splitlines()[1:]}
    assert all(rows[q][1] == ln for q, ln in golden.items()) and len(rows) == len(golden) + len(extra)


def test_streamed_gz_equals_the_plain_file(ctx, ci_libs, tmp_path):
    """A .fastq.gz is inflated in record-aligned pieces on a worker thread and parsed piece by piece (collapse.GzipRecordStream,
    mirge_reads_concat): the reads, their order, the record count and everything behind them are those of the plain file --
    with pieces of a few records, several gzip members, trimming with the count after every modifier, UMIs (not streamed:
    the whole text), and through the CLI with one compressed and one plain sample (digest.py:136-140)."""
    import gzip
    from mirge3_amd import collapse
    reads = synth.make_reads(ci_libs, 40000, seed=12, n_frac=0.01).to_list()
    ad = collapse.ILLUMINA_3P
    rng = np.random.default_rng(3)
    recs = []
    for i, q in enumerate(reads):
        s_ = (q + ad)[:50]
        qual = "".join(chr(33 + int(x)) for x in rng.integers(2, 41, size=len(s_)))
        recs.append(f"@r{i} 1:N:0\n{s_}\n+\n{qual}\n")
    text = "".join(recs).encode()
    plain = tmp_path / "S.fastq"
    plain.write_bytes(text)
    gz = tmp_path / "S.fastq.gz"
    c = len(text) // 2 + 11
    gz.write_bytes(gzip.compress(text[:c], 6) + gzip.compress(text[c:], 6))
    trims = [None, _ffi.MirgeTrim.make(adapter=ad, quality_back=10, count_per_modifier=True),
             _ffi.MirgeTrim.make(adapter=ad, quality_back=20, nextseq=15, trim_n=True, count_per_modifier=False)]
    for trim in trims:
        want, n_want = collapse.parse_sample(ctx, collapse.read_text(str(plain)), 16, trim, None)
        for piece in (700, 50_000, 8 << 20):
            tm = {}
            got, n_got = collapse.parse_sample(ctx, collapse.GzipRecordStream(str(gz), piece_bytes=piece), 16, trim, None, timings=tm)
            assert n_got == n_want == len(reads) and len(got) == len(want)
            assert got.unpack().to_list() == want.unpack().to_list()
            assert np.array_equal(got.group_counts(), want.group_counts())
            assert tm["gz_pieces"] >= 1 and tm["inflate_s"] > 0 and (piece > 1 << 20 or tm["gz_pieces"] > 10)
            u1, u2 = got.collapse(), want.collapse()
            (c1, f1), (c2, f2) = u1.counts(), u2.counts()
            o1, o2 = np.argsort(f1, kind="stable"), np.argsort(f2, kind="stable")
            assert np.array_equal(f1[o1], f2[o2]) and np.array_equal(c1[o1], c2[o2])
            for h in (u1, u2, got):
                h.close()
        # inflated on all host cores and parsed beside (collapse.ParallelGzipStream): here the two members are too small to cut, the
        # route declines at the end and _parse_stream starts over with zlib -- same reads
        got, n_got = collapse.parse_sample(ctx, collapse.ParallelGzipStream(str(gz), piece_bytes=1 << 16), 16, trim, None)
        assert n_got == n_want and got.unpack().to_list() == want.unpack().to_list()
        got.close()
        # the plain text in parts of whole records (collapse.TextRecordStream: how a text of 8 GiB or more is parsed)
        for piece in (900, 200_000):
            tm = {}
            got, n_got = collapse.parse_sample(ctx, collapse.TextRecordStream(np.memmap(plain, dtype=np.uint8, mode="r"), piece), 16, trim, None, timings=tm)
            assert n_got == n_want and got.unpack().to_list() == want.unpack().to_list() and np.array_equal(got.group_counts(), want.group_counts())
            assert tm["gz_pieces"] >= len(text) // piece
            got.close()
        want.close()
    # ... and a member large enough to be cut (3 x the text, one member): pieces parsed while the rest inflates
    big_plain, big_gz = tmp_path / "B.fastq", tmp_path / "B.fastq.gz"
    big_text = b"".join(text.replace(b"@r", b"@%c" % t_) for t_ in b"rst")
    big_plain.write_bytes(big_text)
    big_gz.write_bytes(gzip.compress(big_text, 6))
    assert big_gz.stat().st_size > (3 << 20)
    for trim in trims[:2]:
        want, n_want = collapse.parse_sample(ctx, collapse.read_text(str(big_plain)), 16, trim, None)
        for piece in (1 << 16, 4 << 20):
            tm = {}
            st = collapse.ParallelGzipStream(str(big_gz), piece_bytes=piece)
            got, n_got = collapse.parse_sample(ctx, st, 16, trim, None, timings=tm)
            assert st.job.ok and n_got == n_want == 3 * len(reads) and got.unpack().to_list() == want.unpack().to_list()
            assert np.array_equal(got.group_counts(), want.group_counts()) and tm["gz_pieces"] >= 2
            got.close()
        want.close()
    st = collapse.read_text(str(big_gz), stream=True)
    assert isinstance(st, collapse.ParallelGzipStream)
    st.close()
    umi = _ffi.MirgeUmi.make(4, 2)
    a, na = collapse.parse_sample(ctx, collapse.read_text(str(plain)), 16, trims[1], umi)
    b2, nb2 = collapse.parse_sample(ctx, collapse.ParallelGzipStream(str(gz)), 16, trims[1], umi)  # UMIs: the whole text, here through zlib
    assert na == nb2 and a.unpack().to_list() == b2.unpack().to_list()
    b2.close()
    b, nb = collapse.parse_sample(ctx, collapse.GzipRecordStream(str(gz), piece_bytes=5000), 16, trims[1], umi)
    assert na == nb and a.unpack().to_list() == b.unpack().to_list()
    a.close(); b.close()
    # the CLI: golden case 2, the first sample compressed (pieces of 2 kB), against both samples plain
    case = GoldenCase("case2_two_samples")
    files = _case_fastqs(case, tmp_path)
    with open(files[0], "rb") as fh:
        (tmp_path / "gzrun").mkdir()
        zp = tmp_path / "gzrun" / (os.path.basename(files[0]) + ".gz")
        zp.write_bytes(gzip.compress(fh.read(), 6))
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for tag, fl, env in (("plain", files, {}), ("gz", [str(zp), files[1]], {"MIRGE_GZ_PIECE_BYTES": "2000"}),
                         ("parts", files, {"MIRGE_TEXT_PIECE_BYTES": "3000"})):  # every text of 9 kB or more in parts of 3 kB
        cmd = [sys.executable, "-c", "import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()" % root,
               "-s", ",".join(fl), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", tag, "-shh"]
        r = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
    for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "annotation.report.csv"):
        assert (tmp_path / "gz" / f).read_text() == (tmp_path / "plain" / f).read_text(), f
        assert (tmp_path / "parts" / f).read_text() == (tmp_path / "plain" / f).read_text(), f
        # (the golden report counts the reads below --minimum-length too, which _case_fastqs does not write back)
        assert f == "annotation.report.csv" or (tmp_path / "gz" / f).read_text() == case.text(f), f


def test_last_record_with_an_empty_read(ctx):
    """A file trimmed without a minimum length ends with an empty read now and then: '@r' / '' / '+' / ''.  The blank end of a file
    is stripped before the parse -- but as many of its empty lines as complete the last record are lines (dnaio reads such a record
    too: one more input read, dropped by the length filter); blank lines beyond that are no records, and a file that ends inside a
    record is still refused."""
    seq = "TGAGGTAGTAGGTTGTATAGTT"
    for eol in ("\n", "\r\n"):
        fq = f"@a{eol}{seq}{eol}+{eol}{'I' * 22}{eol}@b{eol}{eol}+{eol}{eol}"
        fa = f">a{eol}{seq}{eol}>b{eol}{eol}"
        for fmt, text in ((1, fq), (2, fa)):
            for extra in ("", eol, eol * 3, " " + eol):
                dr, n_rec = _ffi.DeviceReads.parse(ctx, (text + extra).encode(), fmt, 0)
                assert n_rec == 2 and dr.unpack().to_list() == [seq, ""], (fmt, eol, extra)
                dr.close()
                dr, n_rec = _ffi.DeviceReads.parse(ctx, (text + extra).encode(), fmt, 16)
                assert n_rec == 2 and dr.unpack().to_list() == [seq]
                dr.close()
        # two empty records at the end; an empty record in the middle
        dr, n_rec = _ffi.DeviceReads.parse(ctx, (fq + f"@c{eol}{eol}+{eol}{eol}").encode(), 1, 0)
        assert n_rec == 3 and dr.unpack().to_list() == [seq, "", ""]
        dr.close()
        dr, n_rec = _ffi.DeviceReads.parse(ctx, (f"@b{eol}{eol}+{eol}{eol}" + fq[: fq.index("@b")]).encode(), 1, 0)
        assert n_rec == 2 and dr.unpack().to_list() == ["", seq]
        dr.close()
        # the file ends inside a record: no final line end behind the '+' line -- three lines, refused as before
        with pytest.raises(RuntimeError, match="whole number"):
            _ffi.DeviceReads.parse(ctx, f"@a{eol}{seq}{eol}+{eol}{'I' * 22}{eol}@b{eol}{eol}+".encode(), 1, 0)
        with pytest.raises(RuntimeError, match="whole number"):
            _ffi.DeviceReads.parse(ctx, f"@a{eol}{seq}{eol}+{eol}{'I' * 22}{eol}@b{eol}".encode(), 1, 0)
    # pieces of a stream end the same way
    from mirge3_amd import collapse
    text = ("".join(f"@r{i}\n{seq if i % 3 else ''}\n+\n{'I' * (22 if i % 3 else 0)}\n" for i in range(300))).encode()
    whole, n_whole = collapse.parse_sample(ctx, text, 0, None, None)
    parts, n_parts = collapse.parse_sample(ctx, collapse.TextRecordStream(text, 120), 0, None, None)
    assert n_whole == n_parts == 300 and parts.unpack().to_list() == whole.unpack().to_list() == [seq if i % 3 else "" for i in range(300)]
    whole.close(); parts.close()


@pytest.mark.parametrize("seed", range(6))
def test_cli_route_equals_the_dropin_functions_on_random_samples(seed, tmp_path):
    """The CLI's device-resident route (fastpath.run) and the reference-signature functions (baking -> bwt_align -> DataFrame.to_csv
    -> summarize) are two implementations above the same kernels: on random samples -- 1-3 files of golden reads mutated, cut,
    T-tailed, with random ones mixed in, with or without the 3' adapter, minimum length 12-18, the spike-in library -- every file
    both write must be the same, byte for byte."""
    import subprocess
    import sys
    rng = np.random.default_rng(31000 + seed)
    case = GoldenCase(["case1_single", "case2_two_samples", "case3_spikein", "case4_gff_a2i"][seed % 4])
    S = int(rng.choice([1, 2, 3]))
    ad = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"
    use_ad = bool(rng.random() < 0.5)
    files, names = [], []
    pool = list(case.seqs)
    for s_ in range(S):
        recs = []
        for i in range(int(rng.choice([200, 3000, 20000]))):
            q = pool[int(rng.integers(0, len(pool)))]
            r = rng.random()
            if r < 0.15:
                p_ = int(rng.integers(0, len(q)))
                q = q[:p_] + "ACGTN"[int(rng.integers(0, 5))] + q[p_ + 1:]
            elif r < 0.25:
                q = q[int(rng.integers(0, 3)):len(q) - int(rng.integers(0, 3))]
            elif r < 0.30:
                q = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(10, 45))))
            elif r < 0.33:
                q = q + "TTTT"
            if use_ad:
                q = (q + ad)[:int(rng.integers(30, 76))]
            qual = "".join(chr(int(c)) for c in rng.integers(40, 74, size=len(q)))
            if rng.random() < 0.1 and len(q) > 4:
                qual = qual[:-3] + "###"
            recs.append(f"@r{i}\n{q}\n+\n{qual}\n")
        p = tmp_path / f"S{s_ + 1}.fastq"
        p.write_text("".join(recs))
        files.append(str(p)); names.append(f"S{s_ + 1}")
    min_len = int(rng.choice([16, 16, 18, 12]))
    extra = (["-a", ad] if use_ad else []) + (["-spk"] if case.spike else []) + (["-m", str(min_len)] if min_len != 16 else [])
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, "-c", "import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()" % root,
           "-s", ",".join(files), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", "cli", "-shh"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    wd = tmp_path / "fn"
    wd.mkdir()
    args = SimpleNamespace(threads=1, bowtie_path=None, bowtieVersion="True", quiet=True, bam_out=False, tRNA_frag=False, spikeIn=case.spike,
                           organism_name=ORG, libraries_path=case.libdir, crThreshold="0.1", gff_out=False, isoform_entropy=False, AtoI=False,
                           minimum_length=min_len, adapters=[("back", ad)] if use_ad else None, front=None, uniq_mol_ids=None,
                           quality_cutoff="10", nextseq_trim=None, trim_n=False, cut=[], overlap=3, error_rate=0.12, phred64=33,
                           trim_count="per-modifier", times=1, indels=True, action="trim", match_read_wildcards=False,
                           match_adapter_wildcards=True, umiDedup=False, qiagenumi=False, tcf_out=False)
    df, src, trimmed, uniq = baking(args, files, names, str(wd))
    out = bwt_align(args, df, str(wd), DB)
    mapped, unmapped = out[out.annotFlag.eq(1)], out[out.annotFlag.eq(0)]
    mapped.to_csv(wd / "mapped.csv")
    unmapped.to_csv(wd / "unmapped.csv")
    summarize(args, str(wd), DB, names, mapped, src, trimmed, uniq)
    for f in ("mapped.csv", "unmapped.csv", "annotation.report.csv", "miR.Counts.csv", "miR.RPM.csv"):
        assert (tmp_path / "cli" / f).read_text() == (wd / f).read_text(), f
    assert sum(1 for _ in open(wd / "mapped.csv")) > 100


def test_cli_several_large_gz_samples(ci_libs, tmp_path):
    """Three samples in one invocation, two of them .fastq.gz large enough for the parallel inflater (read ahead on worker threads:
    their inflations overlap each other and the first sample's parse; the text buffer of one serves the next), one plain: every
    output file equals the run on the three plain files, and run.log says nothing went through zlib."""
    import gzip
    import subprocess
    import sys
    case = GoldenCase("case2_two_samples")
    rng = np.random.default_rng(21)
    plain, mixed = [], []
    for k in range(3):
        reads = synth.make_reads(ci_libs, 120000, seed=30 + k, n_frac=0.01).to_list()
        recs = []
        for i, q in enumerate(reads):
            qual = "".join(chr(33 + int(x)) for x in rng.integers(2, 41, size=len(q)))
            recs.append(f"@s{k}_{i}\n{q}\n+\n{qual}\n")
        text = "".join(recs).encode()
        p_ = tmp_path / f"S{k}.fastq"
        p_.write_bytes(text)
        plain.append(str(p_))
        if k != 1:
            (tmp_path / "z").mkdir(exist_ok=True)
            z = tmp_path / "z" / f"S{k}.fastq.gz"
            z.write_bytes(gzip.compress(text, 6))
            assert z.stat().st_size > (2 << 20)
            mixed.append(str(z))
        else:
            mixed.append(str(p_))
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for tag, fl in (("plain", plain), ("gz", mixed)):
        cmd = [sys.executable, "-c", "import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()" % root,
               "-s", ",".join(fl), "-lib", case.libdir, "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", tag, "-shh"]
        r = subprocess.run(cmd, env=dict(os.environ, MIRGE_GZ_TIMING="1"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        if tag == "gz":
            assert r.stderr.count("mirge_gz:") == 2, r.stderr[-2000:]  # both compressed samples took the parallel inflater
    for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "annotation.report.csv"):
        assert (tmp_path / "gz" / f).read_text() == (tmp_path / "plain" / f).read_text(), f


def test_cli_library_cache_next_to_the_index(tmp_path):
    """The one-time conversion kept next to the index (libcache.py): a first CLI run on a fresh copy of golden case 1's
    library directory writes <index>.mirge3amd for every library, a second run loads them (mirge_lib_create_packed: no
    FASTA parse, no packing) and writes the same bytes; touching an index invalidates just that cache."""
    import shutil
    case = GoldenCase("case1_single")
    libdir = tmp_path / "Libs"
    shutil.copytree(case.libdir, libdir)
    fq = _case_fastq_files(case, tmp_path)
    idx = libdir / ORG / "index.Libs"
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cmd = [sys.executable, "-c", "import sys; sys.path.insert(0, %r); import mirge3_amd; from mirge3_amd.cli import main; main()" % root]

    def run(tag):
        r = subprocess.run(cmd + ["-s", ",".join(fq), "-lib", str(libdir), "-on", ORG, "-db", "miRBase", "-o", str(tmp_path), "-dn", tag, "-shh"],
                           env=dict(os.environ, MIRGE_LIB_CACHE="1"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return {f: (tmp_path / tag / f).read_bytes() for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv", "annotation.report.csv")}

    a = run("first")
    caches = sorted(f for f in os.listdir(idx) if f.endswith(".mirge3amd"))
    assert len(caches) == 8, caches
    stamp = {f: os.stat(idx / f).st_mtime_ns for f in caches}
    b = run("second")
    assert a == b and {f: os.stat(idx / f).st_mtime_ns for f in caches} == stamp  # loaded, not rewritten
    for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv"):
        assert a[f].decode() == case.text(f), f
    os.utime(idx / f"{ORG}_rrna.fa", ns=(5, 5))
    c = run("third")
    after = {f: os.stat(idx / f).st_mtime_ns for f in caches}
    assert c == a and [f for f in caches if after[f] != stamp[f]] == [f"{ORG}_rrna.mirge3amd"]


def test_cli_route_edge_inputs(tmp_path):
    """The device-resident CLI route on degenerate inputs: an empty file beside a normal one, a file whose reads are
    all below --minimum-length, and a sample without a single miRNA read under -gff / -ai / -ie -- files are written,
    nothing crashes, counts are what the reads say."""
    from mirge3_amd import fastpath
    case = GoldenCase("case4_gff_a2i")
    work = tmp_path / "out"
    work.mkdir()
    good = tmp_path / "A.fastq"
    with open(good, "w") as fh:
        for seq, row in list(zip(case.seqs, case.counts))[:300]:
            fh.write(f"@r\n{seq}\n+\n{'I' * len(seq)}\n" * min(int(row[1]) + 1, 3))
    empty = tmp_path / "B.fastq"
    empty.write_text("")
    short = tmp_path / "C.fastq"
    short.write_text("@r\nACGTACGT\n+\nIIIIIIII\n" * 5)
    nomir = tmp_path / "D.fastq"
    nomir.write_text("".join(f"@r\n{s}\n+\n{'I' * len(s)}\n" for s in ["GATTACAGATTACAGATTACAGATT", "CCCCCCCCCCCCCCCCCCCCCC"] * 3))
    base = dict(libraries_path=case.libdir, organism_name=ORG, spikeIn=False, quiet=True, minimum_length=16, crThreshold="0.1",
                device=0, threads=1, bowtieVersion="True", phred64=False,
                bowtie_path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fake_bowtie"))
    out = fastpath.run(SimpleNamespace(isoform_entropy=False, **base), [str(good), str(empty), str(short)], ["A", "B", "C"], str(work), "miRBase")
    rep = list(csv.DictReader(open(work / "annotation.report.csv")))
    assert [r["Sample name(s)"] for r in rep] == ["A", "B", "C"]
    assert (int(rep[1]["Total Input Reads"]), int(rep[1]["Trimmed Reads (all)"])) == (0, 0)
    assert (int(rep[2]["Total Input Reads"]), int(rep[2]["Trimmed Reads (all)"])) == (5, 0)
    assert int(rep[0]["Trimmed Reads (all)"]) > 300
    mapped = (work / "mapped.csv").read_text().splitlines()
    assert mapped[0].endswith(",A,B,C") and all(ln.endswith(",0,0") for ln in mapped[1:])
    for h in ("uniq", "res"):
        out["device"][h].close()
    work2 = tmp_path / "out2"
    work2.mkdir()
    out = fastpath.run(SimpleNamespace(isoform_entropy=True, gff_out=True, AtoI=True, **base), [str(nomir)], ["D"], str(work2), "miRBase")
    assert (work2 / "sample_miRge3.gff").read_text().count("\n") == 4  # the four header lines
    assert (work2 / "a2IEditing.report.csv").read_text().count("\n") == 1 and (work2 / "isomirs.csv").exists()
    assert int(list(csv.DictReader(open(work2 / "annotation.report.csv")))[0]["All miRNA Reads"]) == 0


def test_sorted_order_on_the_device(ctx, ci_libs):
    """mirge_collapse_order_sorted == Python's sorted() on the sequences (the index of the reference's outer-joined frame,
    digest.py:243): every read group at once -- reads with N (N sorts between G and T), 32-255-nt reads, reads that are
    prefixes of other reads -- and mirge_collapse_nonzero == the matrix's non-zero counts per column."""
    rng = np.random.default_rng(12)
    base = synth.make_reads(ci_libs, 60000, seed=21, n_frac=0.05, pool=9000).to_list()
    extra = []
    for q in base[:3000]:
        extra += [q[:16], q[:17], q + "A", q + "T", q + "N", q[:20] + "N" + q[21:]]
    longs = ["".join("ACGTN"[int(c)] for c in rng.choice(5, size=int(L), p=[.24, .24, .24, .24, .04])) for L in rng.integers(32, 256, size=400)]
    longs += [longs[0][:100], longs[0][:64], longs[0][:65], longs[1] + "A"][:4]
    samples = [FlatSeqs.from_list(base + extra), FlatSeqs.from_list(extra[::3] + longs), FlatSeqs.from_list(longs[::2] + base[::5])]
    uniq = collapse_samples(ctx, samples)
    seqs = uniq.unpack().to_list()
    got = uniq.sorted_order()
    assert [seqs[i] for i in got] == sorted(seqs) and len(set(seqs)) == len(seqs) > 8000
    cnt, _ = uniq.counts()
    assert uniq.nonzero_per_sample().tolist() == np.count_nonzero(cnt, axis=0).tolist()
    from mirge3_amd.fastpath import row_order
    assert np.array_equal(got, row_order(uniq.unpack(), None, 3))
    uniq.close()


def test_first_appearance_order_on_the_device(ctx):
    """mirge_collapse_order == ranking the first indices on the host (fastpath.row_order), over several read groups."""
    from mirge3_amd.fastpath import row_order
    sl = synth.make_libraries(seed=20260101, scale="small")
    reads = synth.make_reads(sl, 300000, seed=9, n_frac=0.02)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq = raw.collapse()
    cnt, first = uniq.counts()
    got = uniq.first_appearance_order()
    assert np.array_equal(got, row_order(uniq.unpack(), first, 1)) and len(got) == len(uniq) > 1000
    uniq.close(); raw.close()
