#!/usr/bin/env python3
"""Generate tests/golden/<case>/ by running the REFERENCE's own ``bwtAlign`` and ``summarize``
(/root/reference/mirge/libs/manifoldAlign.py:68, summary.py:677) in this container.

Runs only where /root/reference exists (never on the GPU box, never from a test).  What is
committed is data: the inputs (library FASTA, merges CSV, collapsed reads with counts) and the
files the reference wrote (mapped.csv, unmapped.csv, annotation.report.csv, miR.Counts.csv,
miR.RPM.csv).  No reference source is copied.

Recipe:
  * import-time stand-ins for cutadapt / dnaio / xopen / Bio in tests/golden/stubs (the hot
    path never calls into them);
  * `bowtie` / `bowtie-inspect` stand-ins in tests/golden/fake_bowtie, passed through the
    reference's own ``-pbwt`` mechanism (args.bowtie_path); they answer with the oracle's
    brute-force matcher, so these vectors pin the reference's Python around the bowtie process
    boundary (subset rules, T-tail strip, SAM parsing, overwrite rule, count join), not bowtie;
  * the DataFrame handed to bwtAlign is built with the schema of digest.py:237-261 because
    ``baking`` itself needs the real cutadapt/dnaio.

usage: python tests/golden/make_golden.py
"""
import os
import shutil
import sys
import tempfile
from collections import Counter
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "stubs"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

import mirge3_amd  # noqa: E402,F401
from mirge3_amd import synth  # noqa: E402
from mirge3_amd.seqio import Library, FlatSeqs, write_fasta, index_basename  # noqa: E402

from mirge.libs.manifoldAlign import bwtAlign  # noqa: E402  (the reference)
from mirge.libs.summary import summarize  # noqa: E402  (the reference)

ORG, DB = "human", "miRBase"
PASS_COLS = ['exact miRNA', 'hairpin miRNA', 'mature tRNA', 'primary tRNA', 'snoRNA', 'rRNA',
             'ncrna others', 'mRNA', 'isomiR miRNA', 'spike-in']


def edge_reads(sl):
    """Hand-made reads for the boundaries SURVEY.md 8c lists."""
    L = sl.libs
    mir = L["mirna"].seqs.to_list()
    hp = L["hairpin"].seqs.to_list()
    mt = L["mature_trna"].seqs.to_list()
    pre = L["pre_trna"].seqs.to_list()
    mr = L["mrna"].seqs.to_list()
    nc = L["ncrna_others"].seqs.to_list()
    out = []
    out += [mir[0], mir[1][:16], mir[2][1:], mir[3][:-1]]             # exact / substrings
    out += [hp[0][5:30], hp[0][5:31], hp[1][3:31], hp[1][3:32], hp[2][0:40]]  # len 25/26, 28/29
    x = list(hp[3][4:36]); x[29] = "A" if x[29] != "A" else "C"; out.append("".join(x))  # 1 mm beyond seed
    x = list(hp[3][4:36]); x[3] = "A" if x[3] != "A" else "C"; x[30] = "G" if x[30] != "G" else "T"
    out.append("".join(x))                                               # 1 in seed + 1 beyond
    x = list(hp[3][4:36]); x[3] = "A" if x[3] != "A" else "C"; x[9] = "G" if x[9] != "G" else "T"
    out.append("".join(x))                                               # 2 in seed -> no hairpin hit
    out += [mt[0][-20:], mt[1][10:30]]
    x = list(mt[2][5:27]); x[11] = "A" if x[11] != "A" else "C"; out.append("".join(x))  # -v 1
    out += [pre[0][-18:] + "TTTT", pre[1][-15:] + "TTT", pre[2][-17:] + "TT", "T" * 18,
            "ACGTACGTACGTACG" + "TTTTT", pre[3][-1:] + "T" * 17]         # T-tail cases incl. all-T
    out += [mr[0][100:122], mr[1][50:95]]
    x = list(mr[2][10:45]); x[30] = "A" if x[30] != "A" else "C"; x[33] = "G" if x[33] != "G" else "T"
    out.append("".join(x))                                               # mRNA: 2 mm beyond seed
    x = list(mr[2][60:95]); x[5] = "A" if x[5] != "A" else "C"; out.append("".join(x))  # 1 mm in seed -> miss
    # window over the planted reference N run
    nref = next(s for s in nc if "N" in s)
    p = nref.index("N")
    out += [nref[p - 10:p + 12].replace("N", "A"), nref[p - 25:p - 3]]
    # reads with N
    x = list(mir[4]); x[5] = "N"; out.append("".join(x))
    x = list(mt[3][20:44]); x[7] = "N"; out.append("".join(x))
    # isomiRs: 5'/3' shifts and SNVs around a mature in its hairpin
    for i in (5, 6, 7, 8):
        h, off = int(sl.mir_hairpin[i]), int(sl.mir_hairpin_off[i])
        ln = len(mir[i])
        out += [hp[h][off - 1:off + ln + 1], hp[h][off + 1:off + ln + 2], hp[h][off:off + ln] + "A"]
        x = list(hp[h][off:off + ln]); x[10] = "A" if x[10] != "A" else "C"; out.append("".join(x))
        x = list(hp[h][off:off + ln]); x[4] = "A" if x[4] != "A" else "C"; x[15] = "G" if x[15] != "G" else "T"
        out.append("".join(x))
    return [r for r in out if len(r) >= 16]


def special_reads(sl, s):
    """Counts chosen to hit every branch of mirge_can (summary.py:36-42): exact < 2 (zeroed with
    its isomiRs), exact/isomiR ratio below and above -cr 0.1, isomiR-only names (dropped by the
    left merge), for an unmerged miRNA and for both members of a merged family."""
    mir = sl.libs["mirna"].seqs.to_list()
    hp = sl.libs["hairpin"].seqs.to_list()

    def iso(i):  # templated 3' isomiR: one base further into the hairpin, first base dropped
        h, off = int(sl.mir_hairpin[i]), int(sl.mir_hairpin_off[i])
        return hp[h][off + 1:off + len(mir[i]) + 2]

    plan = [  # (miRNA index, exact copies, isomiR copies)
        (30, 2, 30 + s), (31, 1, 5), (32, 0, 6), (33, 3, 29), (2, 1, 4), (3, 2, 40), (34, 2, 19 + 2 * s),
    ]
    out = []
    for i, ne, ni in plan:
        out += [mir[i]] * ne + [iso(i)] * ni
    return out


def write_libs(dirpath, sl, spike):
    idx = os.path.join(dirpath, ORG, "index.Libs")
    ann = os.path.join(dirpath, ORG, "annotation.Libs")
    os.makedirs(idx)
    os.makedirs(ann)
    for key, lib in sl.libs.items():
        write_fasta(os.path.join(idx, index_basename(ORG, key, DB) + ".fa"), lib)
    if spike is not None:
        write_fasta(os.path.join(idx, index_basename(ORG, "spike-in", DB) + ".fa"), spike)
    with open(os.path.join(ann, f"{ORG}_merges_{DB}.csv"), "w") as fh:
        for row in sl.merges:
            fh.write(",".join(row) + "\n")


def build_frame(sample_dicts, base_names):
    """Schema of digest.py:237-261."""
    complete_set = pd.DataFrame()
    for name, d in zip(base_names, sample_dicts):
        collapsed_df = pd.DataFrame(list(d.items()), columns=['Sequence', name])
        collapsed_df.set_index('Sequence', inplace=True)
        if len(base_names) == 1:
            complete_set = collapsed_df
        else:
            complete_set = complete_set.join(collapsed_df, how='outer')
        complete_set = complete_set.fillna(0).astype(int)
    complete_set = complete_set.assign(**dict.fromkeys(PASS_COLS, ''))
    complete_set = complete_set.assign(**dict.fromkeys(['annotFlag'], '0'))
    complete_set = complete_set.reindex(columns=['annotFlag'] + PASS_COLS + base_names)
    return complete_set.astype({"annotFlag": int})


def run_case(case, seed, n_raw, n_samples, spike_in):
    out_dir = os.path.join(HERE, case)
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(os.path.join(out_dir, "libs"))
    sl = synth.make_libraries(seed=seed, scale="tiny")
    # a few miRNAs nobody reads (zero rows of miR.Counts.csv / miR.RPM.csv)
    rng0 = np.random.Generator(np.random.PCG64(seed + 99))
    mir = sl.libs["mirna"]
    silent = ["".join("ACGT"[int(c)] for c in rng0.integers(0, 4, size=21)) for _ in range(7)]
    n_gen = len(mir)  # reads are only drawn from the first n_gen miRNAs
    sl.libs["mirna"] = Library(mir.names + [f"hsa-miR-silent-{i + 1}" for i in range(7)],
                               FlatSeqs.from_list(mir.seqs.to_list() + silent))
    sl_gen = synth.SynthLibs(dict(sl.libs, mirna=mir), sl.merges, sl.mir_hairpin, sl.mir_hairpin_off)
    spike = None
    if spike_in:
        rng = np.random.Generator(np.random.PCG64(seed + 5))
        sp = ["".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=22)) for _ in range(6)]
        spike = Library([f"spike-{i + 1}" for i in range(6)], FlatSeqs.from_list(sp))
    tmp = tempfile.mkdtemp(prefix="mirge_golden_")
    libdir = os.path.join(tmp, "Libs")
    write_libs(libdir, sl, spike)
    shutil.copytree(os.path.join(libdir, ORG), os.path.join(out_dir, "libs", ORG))
    base_names = [f"S{i + 1}" for i in range(n_samples)]
    dicts, src, trimmed, uniq = [], {}, {}, {}
    edges = edge_reads(sl)
    for s, name in enumerate(base_names):
        raw = synth.make_reads(sl_gen, n_raw, seed=seed * 10 + s).to_list()
        raw += edges[s::n_samples] * 2 + edges[: len(edges) // 2]
        if spike is not None:
            raw += spike.seqs.to_list() * 3 + [spike.seqs.get(0)[:18]]
        raw += special_reads(sl, s)
        d = dict(Counter(raw))  # collapse rule of digest.py:158-163 (insertion order = first seen)
        dicts.append(d)
        src[name] = len(raw) + 17  # pretend 17 reads were dropped by trimming
        trimmed[name] = sum(d.values())
        uniq[name] = len(d)
    df = build_frame(dicts, base_names)
    work = os.path.join(tmp, "work")
    os.makedirs(work)
    args = SimpleNamespace(threads=2, bowtie_path=os.path.join(HERE, "fake_bowtie"), bowtieVersion="True",
                           quiet=True, bam_out=False, tRNA_frag=False, spikeIn=bool(spike_in),
                           organism_name=ORG, libraries_path=libdir, crThreshold="0.1", gff_out=False,
                           isoform_entropy=True, AtoI=False)  # -ie: isomirs.csv, isomirs.samples.csv (summary.py:906-1032)
    # inputs
    df[base_names].to_csv(os.path.join(out_dir, "collapsed_input.csv"))
    with open(os.path.join(out_dir, "counters.csv"), "w") as fh:
        fh.write("sample,total_input,trimmed_all,trimmed_unique\n")
        for nme in base_names:
            fh.write(f"{nme},{src[nme]},{trimmed[nme]},{uniq[nme]}\n")
    # the reference, exactly as mirge/__main__.py:157-173 drives it
    df = bwtAlign(args, df, work, DB)
    pdMapped = df[df.annotFlag.eq(1)]
    pdUnmapped = df[df.annotFlag.eq(0)]
    summarize(args, work, DB, base_names, pdMapped, src, trimmed, uniq)
    pdMapped.to_csv(os.path.join(work, "mapped.csv"))
    pdUnmapped.to_csv(os.path.join(work, "unmapped.csv"))
    for f in ("mapped.csv", "unmapped.csv", "annotation.report.csv", "miR.Counts.csv", "miR.RPM.csv", "isomirs.csv",
              "isomirs.samples.csv"):
        shutil.copy(os.path.join(work, f), os.path.join(out_dir, f))
    shutil.rmtree(tmp)
    print(case, "rows", len(df), "mapped", len(pdMapped), "->", out_dir)


def run_umi_case():
    """UMI handling (digest.py:164-205,305-315,358-365): the reference's own ``UMIParser`` answers for
    the slicing (incl. its ``s[-0:]`` behaviour); the dict loops around it are the 40 lines of
    ``baking`` that cannot run without cutadapt/dnaio, replayed here with plain dicts."""
    import json
    from mirge.libs.digest import UMIParser  # the reference
    out_dir = os.path.join(HERE, "umi")
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    sl = synth.make_libraries(seed=21, scale="tiny")
    rng = np.random.Generator(np.random.PCG64(77))
    inserts = synth.make_reads(sl, 300, seed=5).to_list()
    inserts = inserts + inserts[:40] * 3 + [inserts[7][:15], inserts[8][:16], "ACGTACGTACGTACGT"]
    tags = ["".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=8)) for _ in range(12)]
    raw = []
    for ins in inserts:
        for _ in range(int(rng.integers(1, 4))):
            t = tags[int(rng.integers(0, len(tags)))]
            raw.append(t[:4] + ins + t[4:])
    raw += [raw[3]] * 5 + [raw[10]] * 2 + ["ACGT" * 5 + "N" + "ACGT" * 3]
    order = rng.permutation(len(raw))
    raw = [raw[i] for i in order]
    with open(os.path.join(out_dir, "reads.txt"), "w") as fh:
        fh.write("\n".join(raw) + "\n")
    cases = []
    for f, b in ((4, 4), (0, 4), (4, 0), (3, 30), (0, 0)):
        min_len = 16
        parser = [[s_, *UMIParser(s_, f, b)] for s_ in raw[:25] + ["ACGT", "", "ACGTACG"]]
        completeDict = {}
        for s_ in raw:  # cutadapt() worker, digest.py:358-365, + merge :158-163
            if len(UMIParser(s_, f, b)[0]) >= min_len:
                completeDict[s_] = completeDict.get(s_, 0) + 1
        res = {}
        for dedup in (False, True):  # digest.py:164-205
            d, trimmed, lines = {}, 0, ["UMISeq,transcriptSeq,UMICounts\n"]
            for s_, c in completeDict.items():
                pureSeq, cutumiSeq = UMIParser(s_, f, b)
                if len(pureSeq) >= min_len:
                    if dedup:
                        lines.append(str(cutumiSeq) + "," + str(pureSeq) + "," + str(c) + "\n")
                    d[pureSeq] = d.get(pureSeq, 0) + (1 if dedup else c)
                    trimmed += 1 if dedup else c
            res["udd" if dedup else "umi"] = {"dict": list(d.items()), "trimmed": trimmed, "unique": len(d),
                                              "umiCounts_csv": "".join(lines) if dedup else None}
        cases.append({"front": f, "back": b, "min_len": min_len, "total_input": len(raw), "parser": parser, **res})
    with open(os.path.join(out_dir, "umi_cases.json"), "w") as fh:
        json.dump(cases, fh, indent=0)
    print("umi", len(raw), "reads ->", out_dir)


if __name__ == "__main__":
    run_umi_case()
    run_case("case1_single", seed=11, n_raw=1200, n_samples=1, spike_in=False)
    run_case("case2_two_samples", seed=12, n_raw=900, n_samples=2, spike_in=False)
    run_case("case3_spikein", seed=13, n_raw=600, n_samples=2, spike_in=True)
