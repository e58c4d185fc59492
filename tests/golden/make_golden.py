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
from pathlib import Path
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


def run_case(case, seed, n_raw, n_samples, spike_in, cr="0.1"):
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
                           organism_name=ORG, libraries_path=libdir, crThreshold=cr, gff_out=False,
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
    for f in ("mapped.csv", "unmapped.csv", "annotation.report.csv", "annotation.report.html", "miR.Counts.csv", "miR.RPM.csv", "isomirs.csv",
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


# ---------------------------------------------------------------------------------------------------
# case4: -gff (create_gff, summary.py:48-606) and -ai (a2i_editing, mirge2_tRF_a2i.py:979-1432) through the
# reference's own summarize(), plus direct calls of judgeAllign / A2IEditing / mismatchCountAnalysis / UID.
# ---------------------------------------------------------------------------------------------------
WORKED = [  # the three worked examples in the comments of summary.py:249-283 (miRNA name, canonical, read)
    ("hsa-miR-548al", "AACGGCAATGACTTTTGTACCA", "AAACGGCAATGACTTTTGTACT"),
    ("hsa-miR-335-3p", "TTTTTCATTATTGCTCCTGACC", "TTTTTCATTATTGGTCCTGCTC"),
    ("hsa-miR-9999-5p", "AAACCGTTACCATTACTGAGTT", "AAACCGTTTCCATTACTGGGG"),
]


def _sub(seq, pos, to=None):
    x = list(seq)
    x[pos] = to if to is not None else ("A" if x[pos] != "A" else "C")
    return "".join(x)


def gff_a2i_libs(seed):
    """tiny synthetic set + what -gff / -ai read beside the indexes: names with '.SNP' variants, the worked
    examples, fasta.Libs/<org>_mature_<db>.fa, fasta.Libs/<org>_mirna_SNP_pseudo_<db>.fa,
    annotation.Libs/<org>_<db>.gff3 (miRBase layout), annotation.Libs/<org>_miRNAs_in_repetitive_element_<db>.csv
    and a small <org>_genome.fa for the stand-in bowtie."""
    sl = synth.make_libraries(seed=seed, scale="tiny")
    rng = np.random.Generator(np.random.PCG64(seed + 4242))
    mir, hp = sl.libs["mirna"], sl.libs["hairpin"]
    names = [n.replace("-SNP", ".SNP") for n in mir.names]  # real libraries name SNP variants '<name>.SNP<n>'
    seqs = mir.seqs.to_list()
    hps = hp.seqs.to_list()
    hp_names = list(hp.names)
    mir_hp, mir_off = list(sl.mir_hairpin), list(sl.mir_hairpin_off)
    for nm, can, _ in WORKED:
        lead = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=12))
        trail = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=40))
        names.append(nm); seqs.append(can)
        hp_names.append(nm.replace("miR", "mir").rsplit("-", 1)[0] if nm[-2:] in ("3p", "5p") else nm.replace("miR", "mir"))
        hps.append(lead + can + trail)
        mir_hp.append(len(hps) - 1); mir_off.append(len(lead))
    # merged families as they are in the real libraries: members that differ by a base or two (the synthetic
    # generator pairs unrelated sequences).  Six families; member b becomes member a with one 3'-side change.
    fams = []
    for r in sl.merges[:6]:
        a, b = names.index(r[1]), names.index(r[2])
        q = seqs[a]
        seqs[b] = _sub(q, len(q) - 2 - (a % 3))
        h, o = mir_hp[b], mir_off[b]
        hq = hps[h]
        o = min(o, len(hq) - len(q) - 2)
        hps[h] = hq[:o] + seqs[b] + hq[o + len(q):]
        mir_off[b] = o
        fams.append(r)
    # matures that share a hairpin may have been overwritten above or by the generator: re-stamping is not
    # attempted -- create_gff simply finds no precursor position for those (start 0), as in the reference
    sl.libs["mirna"] = Library(names, FlatSeqs.from_list(seqs))
    sl.libs["hairpin"] = Library(hp_names, FlatSeqs.from_list(hps))
    sl.mir_hairpin, sl.mir_hairpin_off = np.asarray(mir_hp), np.asarray(mir_off)
    sl.merges = fams
    return sl


def write_gff_a2i_aux(libdir, sl, seed):
    rng = np.random.Generator(np.random.PCG64(seed + 777))
    fa = os.path.join(libdir, ORG, "fasta.Libs")
    ann = os.path.join(libdir, ORG, "annotation.Libs")
    idx = os.path.join(libdir, ORG, "index.Libs")
    os.makedirs(fa, exist_ok=True)
    mir, hp = sl.libs["mirna"], sl.libs["hairpin"]
    mseq, hseq = mir.seqs.to_list(), hp.seqs.to_list()
    merged_of = {m: r[0] for r in sl.merges for m in r[1:]}
    with open(os.path.join(fa, f"{ORG}_mature_{DB}.fa"), "w") as fh:
        for n, q in zip(mir.names, mseq):
            if ".SNP" not in n:
                fh.write(f">{n}\n{q}\n")
    with open(os.path.join(fa, f"{ORG}_mirna_SNP_pseudo_{DB}.fa"), "w") as fh:
        done = set()
        for n, q in zip(mir.names, mseq):
            for key in (n, merged_of.get(n)):
                if key and key not in done:
                    done.add(key)
                    fh.write(f">{key}\n{q}\n")
    # miRBase-style GFF3: a primary transcript line, then the matures that derive from it; a few on the
    # minus strand; every 11th mature is left out (create_gff then drops its reads: KeyError -> pass)
    members = {}
    for i, n in enumerate(mir.names):
        if ".SNP" in n or i % 11 == 10:
            continue
        members.setdefault(int(sl.mir_hairpin[i]), []).append(i)
    with open(os.path.join(ann, f"{ORG}_{DB}.gff3"), "w") as fh:
        fh.write("##gff-version 3\n# synthetic annotation in the layout of miRBase's hsa.gff3\n")
        pos = 10000
        for h, hn in enumerate(hp.names):
            strand = "-" if h % 5 == 3 else "+"
            chrom = f"chr{1 + h % 3}"
            fh.write(f"{chrom}\t.\tmiRNA_primary_transcript\t{pos}\t{pos + len(hseq[h]) - 1}\t.\t{strand}\t.\t"
                     f"ID=MI{h:07d};Alias=MI{h:07d};Name={hn}\n")
            for i in members.get(h, []):
                o = int(sl.mir_hairpin_off[i])
                fh.write(f"{chrom}\t.\tmiRNA\t{pos + o}\t{pos + o + len(mseq[i]) - 1}\t.\t{strand}\t.\t"
                         f"ID=MIMAT{i:07d};Alias=MIMAT{i:07d};Name={mir.names[i]};Derives_from=MI{h:07d}\n")
            pos += 5000
    with open(os.path.join(ann, f"{ORG}_miRNAs_in_repetitive_element_{DB}.csv"), "w") as fh:
        fh.write(f"{mir.names[6]},LINE\n{mir.names[40]},SINE\n")
    # genome for the stand-in bowtie: three random chromosomes with most hairpins written in once, a few twice
    chroms = []
    for c in range(3):
        g = ACGT_LIST(rng, 6000)
        chroms.append(g)
    for h, q in enumerate(hseq):
        if h % 7 == 6:
            continue  # not in the genome: its reads have no alignment and are never 'retained'
        for rep in range(2 if h % 9 == 4 else 1):
            c = (h + rep) % 3
            p = int(rng.integers(0, 6000 - len(q)))
            chroms[c][p:p + len(q)] = list(q)
    with open(os.path.join(idx, f"{ORG}_genome.fa"), "w") as fh:
        for c, g in enumerate(chroms):
            fh.write(f">chr{c + 1}\n{''.join(g)}\n")


def ACGT_LIST(rng, n):
    return ["ACGT"[int(c)] for c in rng.integers(0, 4, size=n)]


def a2i_reads(sl, s):
    """reads that drive a2i_editing: canonical reads with counts, A>G edited reads at positions outside the last
    five, the same with 3' additions / 5' shifts, counts that sit on both sides of the RPM gates"""
    mir = sl.libs["mirna"]
    mseq = mir.seqs.to_list()
    out = {}
    k = 0
    for i in range(20, 44):
        q = mseq[i]
        if ".SNP" in mir.names[i]:
            continue
        apos = [p for p in range(1, len(q) - 6) if q[p] == "A"]
        if not apos:
            continue
        p = apos[(i + s) % len(apos)]
        out[q] = out.get(q, 0) + 150 + 37 * k + 11 * s
        e = _sub(q, p, "G")
        out[e] = out.get(e, 0) + (1 + (k * 7) % 40) * (1 if (k + s) % 4 else 0)
        if k % 3 == 0:
            out[e + "A"] = out.get(e + "A", 0) + 2 + k % 5
        if k % 4 == 1 and len(apos) > 1:
            e2 = _sub(q, apos[-1], "G")
            out[e2[1:]] = out.get(e2[1:], 0) + 3 + s
        if k % 5 == 2:
            out[_sub(q, p, "C")] = 4
            out[_sub(q, min(p + 3, len(q) - 7), "T" if q[min(p + 3, len(q) - 7)] != "T" else "G")] = 6
        k += 1
    return {q: c for q, c in out.items() if c > 0 and len(q) >= 16}


def gff_reads(sl):
    """isomiR shapes for create_gff: templated / non-templated additions at both ends, trimmed ends, SNVs in every
    region (seed, central offset, central, supplementary, 3'), combinations, an N read, the worked examples"""
    mir, hp = sl.libs["mirna"], sl.libs["hairpin"]
    mseq, hseq = mir.seqs.to_list(), hp.seqs.to_list()
    out = [r for _, _, r in WORKED] + [c for _, c, _ in WORKED]
    for i in range(0, 20):
        if ".SNP" in mir.names[i]:
            continue
        h, o = int(sl.mir_hairpin[i]), int(sl.mir_hairpin_off[i])
        q = mseq[i]
        L = len(q)
        if hseq[h][o:o + L] != q:
            continue  # overwritten by a later mature in the same hairpin
        pre = hseq[h]
        out += [pre[o - 1:o + L], pre[o + 1:o + L], pre[o:o + L + 1], pre[o:o + L + 2], pre[o:o + L - 1], pre[o:o + L - 2],
                pre[o - 1:o + L + 1], pre[o + 1:o + L + 2], pre[o + 1:o + L - 1]]
        nt5 = "A" if pre[o - 1] != "A" else "C"
        nt3 = "A" if pre[o + L] != "A" else "T"
        out += [nt5 + q, q + nt3, q + nt3 + nt3, q + pre[o + L] + nt3, nt5 + q + nt3]
        for p in (2, 5, 7, 9, 12, 15, L - 4):
            out.append(_sub(q, p))
        out += [_sub(_sub(q, 3), 14), _sub(q, 10)[1:], _sub(q, 8) + nt3, _sub(q, 6)[:-1]]
    x = list(mseq[1]); x[9] = "N"; out.append("".join(x))
    return [r for r in out if len(r) >= 16]


def run_gff_a2i_case(case="case4_gff_a2i", seed=14):
    import json
    import filecmp  # noqa: F401
    from mirge.libs import mirge2_tRF_a2i as ref_a2i  # the reference
    from mirge.libs.miRgeEssential import UID  # the reference
    from Bio import pairwise2 as standin
    out_dir = os.path.join(HERE, case)
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(os.path.join(out_dir, "libs"))
    sl = gff_a2i_libs(seed)
    tmp = tempfile.mkdtemp(prefix="mirge_golden_")
    libdir = os.path.join(tmp, "Libs")
    write_libs(libdir, sl, None)
    write_gff_a2i_aux(libdir, sl, seed)
    shutil.copytree(os.path.join(libdir, ORG), os.path.join(out_dir, "libs", ORG))
    base_names = ["S1", "S2"]
    # reads are drawn from the miRNAs as the synthetic generator knows them
    dicts, src, trimmed, uniq = [], {}, {}, {}
    for s, name in enumerate(base_names):
        raw = synth.make_reads(sl, 700, seed=seed * 10 + s).to_list()
        raw += gff_reads(sl)[s::2] * (2 + s) + gff_reads(sl)[1 - s::2]
        d = dict(Counter(raw))
        for q, c in a2i_reads(sl, s).items():
            d[q] = d.get(q, 0) + c
        # one very abundant miRNA lifts 'Filtered miRNA Reads' above 10^6 in S1 only, so that the RPM >= 1 gates
        # of a2i_editing (:1011, :1130-1134) cut at count 2 there and never in S2
        if s == 0:
            d[sl.libs["mirna"].seqs.get(45)] = 1_900_000
        dicts.append(d)
        src[name] = sum(d.values()) + 5
        trimmed[name] = sum(d.values())
        uniq[name] = len(d)
    df = build_frame(dicts, base_names)
    work = os.path.join(tmp, "work")
    os.makedirs(work)
    args = SimpleNamespace(threads=2, bowtie_path=os.path.join(HERE, "fake_bowtie"), bowtieVersion="True",
                           quiet=True, bam_out=False, tRNA_frag=False, spikeIn=False, organism_name=ORG,
                           libraries_path=libdir, crThreshold="0.1", gff_out=True, isoform_entropy=False, AtoI=True,
                           phred64=False)
    df[base_names].to_csv(os.path.join(out_dir, "collapsed_input.csv"))
    with open(os.path.join(out_dir, "counters.csv"), "w") as fh:
        fh.write("sample,total_input,trimmed_all,trimmed_unique\n")
        for nme in base_names:
            fh.write(f"{nme},{src[nme]},{trimmed[nme]},{uniq[nme]}\n")
    # the stand-in's tie rule must never decide a fixture: record every (target, read) pair the reference aligns
    seen_pairs = []
    real_localms = standin.align.localms

    def spy(a, b, *rest):
        seen_pairs.append((a, b))
        return real_localms(a, b, *rest)
    standin.align.localms = spy
    df = bwtAlign(args, df, work, DB)
    pdMapped = df[df.annotFlag.eq(1)]
    pdUnmapped = df[df.annotFlag.eq(0)]
    os.environ["LC_ALL"] = "C"  # a2i_editing sorts its report with the shell's `sort`
    # bowtie's two genome runs are answered by the stand-in; keep what it said (the injectable predicate of the build)
    summarize(args, Path(work), DB, base_names, pdMapped, src, trimmed, uniq)
    standin.align.localms = real_localms
    ambiguous = [(a, b) for a, b in set(seen_pairs) if len(standin.align.diagonals_at_best(a, b, 2, -1, -20, -20)) != 1]
    assert not ambiguous, f"best score on several diagonals (stand-in tie rule would decide): {ambiguous[:3]}"
    pdMapped.to_csv(os.path.join(work, "mapped.csv"))
    pdUnmapped.to_csv(os.path.join(work, "unmapped.csv"))
    for f in ("mapped.csv", "unmapped.csv", "annotation.report.csv", "annotation.report.html", "miR.Counts.csv", "miR.RPM.csv",
              "sample_miRge3.gff", "a2IEditing.report.csv", "a2IEditing.report.newform.csv", "a2IEditing.detail.txt"):
        shutil.copy(os.path.join(work, f), os.path.join(out_dir, f))
    # the genome filter's answer (retainedSeqDic, :1074-1096; removedSeqList, :1310-1316), recomputed with the same
    # stand-in so that the build can be handed it as its injectable predicate
    import subprocess
    fb = os.path.join(HERE, "fake_bowtie", "bowtie")
    genome = os.path.join(libdir, ORG, "index.Libs", f"{ORG}_genome")
    mapped_mi = pdMapped[pdMapped['exact miRNA'].astype(bool) | pdMapped['isomiR miRNA'].astype(bool)]
    fa_all = os.path.join(tmp, "all_mirna_reads.fa")
    with open(fa_all, "w") as fh:
        for q in mapped_mi.index:
            fh.write(f">{q}\n{q}\n")
    o = subprocess.run([sys.executable, fb, "--threads", "1", genome, "-n", "1", "-f", "-a", "-3", "2", fa_all],
                       check=True, stdout=subprocess.PIPE, text=True).stdout
    content = {}
    for row in o.split("\n"):
        f = row.split("\t")
        if f != ['']:
            content.setdefault(f[0], []).append(f[-1].count(":"))
    retained = sorted(q for q, c in content.items() if len(c) == 1 or c.count(min(c)) == 1)
    with open(os.path.join(out_dir, "genome_retained.txt"), "w") as fh:
        fh.write("# reads of the miRNA classes the stand-in genome run retains (unique best alignment, "
                 "mirge2_tRF_a2i.py:1085-1096)\n" + "\n".join(retained) + "\n")
    # ---- direct calls of the reference's functions (row a16)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    mseq = sl.libs["mirna"].seqs.to_list()
    groups = []
    for gi in range(40):
        target = mseq[int(rng.integers(0, len(mseq)))]
        reads, counts = [], []
        for _ in range(int(rng.integers(2, 12))):
            L = len(target)
            d5, d3 = int(rng.integers(-2, 3)), int(rng.integers(-3, 4))
            body = target[max(d5, 0):L + min(d3, 0)]
            q = "".join(ACGT_LIST(rng, max(-d5, 0))) + body + "".join(ACGT_LIST(rng, max(d3, 0)))
            for _k in range(int(rng.choice([0, 0, 1, 1, 2, 3]))):
                p = int(rng.integers(0, len(q)))
                q = _sub(q, p, "ACGT"[int(rng.integers(0, 4))])
            if rng.random() < 0.3:
                apos = [p for p in range(len(q)) if q[p] == "A"]
                if apos:
                    q = _sub(q, apos[int(rng.integers(0, len(apos)))], "G")
            if len(q) < 14 or q in reads:
                continue
            if len(standin.align.diagonals_at_best(target, q, 2, -1, -20, -20)) != 1:
                continue
            reads.append(q); counts.append(int(rng.integers(1, 500)))
        if len(reads) < 2:
            continue
        retained_g = {q: True for q in reads if rng.random() < 0.7}
        aligned, states = ref_a2i.align2TargetSeq(target, reads)
        with open(os.devnull, "w") as devnull:
            kept, plist, pcount, pratio, ppval, count_true, seq_true, canon = ref_a2i.A2IEditing(
                target, reads, counts, "x", devnull, retained_g, 'A', 'G')
        mm = ref_a2i.mismatchCountAnalysis(target, reads, counts, retained_g)
        groups.append(dict(target=target, reads=reads, counts=counts, retained=sorted(retained_g),
                           aligned=aligned, states=[bool(x) for x in states],
                           a2i=dict(kept=kept, positions=plist, count={str(k): v for k, v in pcount.items()},
                                    ratio={str(k): v for k, v in pratio.items()},
                                    pvalue={str(k): float(v) for k, v in ppval.items()}, countSumTrue=count_true,
                                    seqCountTrue=seq_true, canonicalSeqCount=canon),
                           mismatch_census=[list(t) for t in mm]))
    uid = [[q, kind, UID(q, kind)] for q, kind in
           [(mseq[0], "ref"), (mseq[1], "iso"), ("ACGTACGTACGTACGTACGTAC", "iso"), ("TTTTT", "ref"), ("GATTACA", "iso"),
            (mseq[5] + "AT", "iso"), ("A", "ref")]]
    with open(os.path.join(out_dir, "a2i_direct.json"), "w") as fh:
        json.dump(dict(note="outputs of the reference's align2TargetSeq/judgeAllign, A2IEditing, mismatchCountAnalysis "
                            "(mirge2_tRF_a2i.py:246-518) and miRgeEssential.UID; pairwise2 = tests/golden/stubs stand-in, "
                            "only pairs whose best score lies on one diagonal", groups=groups, uid=uid), fh, indent=0)
    shutil.rmtree(tmp)
    print(case, "rows", len(df), "mapped", len(pdMapped), "a2i groups", len(groups), "->", out_dir)


def run_validate_case():
    """miRgeEssential.validate_files (:102-127), the reference's own function, on a directory of file names: which it keeps and
    what it calls the samples -> tests/golden/validate_files.json"""
    import json
    from mirge.libs.miRgeEssential import validate_files  # the reference
    present = ["a.fastq", "b.fq.gz", "c.fastq.gz", "d.fq", "e.trimmed.fastq", "f.tar.fastq.gz", "g.v2.fq", "h.1.2.fq.gz", "my.sample v1.fastq.gz",
               "notes.txt", "lib.fa", "i.fastq.bak", "j.FASTQ", "k.fq.gz.gz", "l.xfastq", "noext", "m.gz", ".fastq", "n.fastq.gz.tmp", "o.fq.GZ",
               "p..fastq", "q.fastq.fastq", "r.fastq.fq", "s.fq.fastq.gz"]
    missing = ["zz.fastq", "yy.fq.gz", "xx.txt"]
    tmp = tempfile.mkdtemp(prefix="mirge_validate_")
    for n in present:
        open(os.path.join(tmp, n), "w").close()
    given = sorted(present) + missing + ["a.fastq"]  # (a name given twice is kept twice)
    full, names = validate_files(SimpleNamespace(quiet=True), [os.path.join(tmp, n) for n in given], os.path.join(tmp, "run.log"), [], [])
    out = {"present": present, "missing": missing, "given": given, "kept": [os.path.basename(f) for f in full], "base_names": names,
           "made_by": "tests/golden/make_golden.py::run_validate_case -- mirge.libs.miRgeEssential.validate_files itself"}
    with open(os.path.join(HERE, "validate_files.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    shutil.rmtree(tmp)
    print("validate_files:", len(full), "of", len(given), "kept")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "validate":
        run_validate_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "case4":
        run_gff_a2i_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "case6":  # (round 4: the -gff / -ai case again, other libraries and reads)
        run_gff_a2i_case(case="case6_gff_a2i", seed=27)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "case7":  # (round 4: -ex 0.4, the canonical-ratio cut well above its default)
        run_case("case7_two_samples_cr0.4", seed=33, n_raw=1400, n_samples=2, spike_in=False, cr="0.4")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "case5":  # (added in round 4: three samples, spike-in library, another seed and depth)
        run_case("case5_three_samples_spikein", seed=21, n_raw=1500, n_samples=3, spike_in=True)
        sys.exit(0)
    run_umi_case()
    run_validate_case()
    run_case("case1_single", seed=11, n_raw=1200, n_samples=1, spike_in=False)
    run_case("case2_two_samples", seed=12, n_raw=900, n_samples=2, spike_in=False)
    run_case("case3_spikein", seed=13, n_raw=600, n_samples=2, spike_in=True)
    run_case("case5_three_samples_spikein", seed=21, n_raw=1500, n_samples=3, spike_in=True)
    run_case("case7_two_samples_cr0.4", seed=33, n_raw=1400, n_samples=2, spike_in=False, cr="0.4")
    run_gff_a2i_case()
    run_gff_a2i_case(case="case6_gff_a2i", seed=27)
