"""Per-position variant tally and the A-to-I editing report (BASELINE config 5) -- host
mirror of ``a2i_editing`` (``mirge/libs/mirge2_tRF_a2i.py:979-1432``, called from ``summary.py:1034-1057`` under ``-ai``).

GPU (``mirge_variant_tally`` / ``k_tally``): for every read the cascade annotated to a miRNA -- membership in its
family's list (:988-1016), the alignment to the family's canonical sequence (``pairwise2.align.localms``, :254), the
acceptance rule ``judgeAllign`` (:298-332), and every count the reference derives per (miRNA, sample): list length,
accepted-and-retained sequences / reads / canonical reads (``A2IEditing``, :335-419) and the twelve base changes per
position in their raw / accepted / accepted-and-retained variants (``mismatchCountAnalysis``, :424-518).

Host (this module): what is left is arithmetic on tables with one row per miRNA -- the RPM gates (:1130-1137), ratio,
binomial p-value (:405-413), Benjamini-Hochberg (:1175-1183) -- and the files: ``a2IEditing.report.csv``,
``a2IEditing.report.newform.csv``, ``a2IEditing.detail.txt``.  The reference deletes ``mismatchCount.csv`` right after
writing it (:1333-1350); it is kept here.

The genome filter stays a predicate: the reference shells out to ``bowtie`` against ``<org>_genome``
twice (:1056-1096, :1297-1316); ``BowtieGenome`` does exactly that through ``args.bowtie_path`` / PATH, ``ListedGenome``
answers from two files of sequences, and any object with ``unique_best(seqs)`` / ``aligned(seqs)`` can be passed in.

Parity: pinned against the reference's own functions and files (``tests/golden/case4_gff_a2i``) except for the
aligner itself, which is Biopython's (absent here; restated, see ``tests/golden/stubs/Bio/pairwise2.py``).
"""
from __future__ import annotations

import math
import os
import shlex
import subprocess
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import _ffi
from .cascade import Cascade, EXACT_PASS, ISO_PASS
from .seqio import FlatSeqs

P_MISMATCH = 0.001   # mirge2_tRF_a2i.py:336
PCT_CUTOFF = 2.0     # a2IPercentageCutoff, :980
A, G = 0, 2
CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
BASE_PAIRS = [('A', 'G'), ('A', 'C'), ('A', 'T'), ('T', 'G'), ('T', 'A'), ('T', 'C'), ('C', 'G'), ('C', 'A'), ('C', 'T'),
              ('G', 'A'), ('G', 'C'), ('G', 'T')]  # column order of mismatchCount.csv (:1336-1338)
# sample labels the reference hard-codes for its own data sets (:1355-1364)
SAMPLE_LABELS = {'SRR837842': 'Colon 1', 'SRR837839': 'Colon 2', 'SRR5127219': 'Colon cell', 'SRR1646473': 'Colon cancer 1',
                 'SRR1646493': 'Colon cancer 2', 'SRR1917324': 'DKO1', 'SRR1917336': 'DLD1', 'SRR1917329': 'DKS8',
                 'SRR567638': 'Placenta 2'}


# ---------------------------------------------------------------------------------------------------------------
# genome predicates
# ---------------------------------------------------------------------------------------------------------------
class BowtieGenome:
    """The reference's own two genome runs: ``bowtie <org>_genome -n 1 -f -a -3 2`` (a sequence is retained when it has
    one alignment, or one alignment with the fewest mismatches; :1056-1096) and ``-n 0 -f -a -3 2`` (:1297-1316)."""

    def __init__(self, args, workDir):
        self.workDir = Path(workDir)
        # (the reference builds one string for a shell, mirge2_tRF_a2i.py:1056-1061; the same words as an argument list)
        self.argv = [str(Path(args.bowtie_path) / "bowtie") if getattr(args, "bowtie_path", None) else "bowtie",
                     "--threads", str(getattr(args, "threads", 1) or 1)]
        if getattr(args, "bowtieVersion", "True") == "False":
            self.argv.append("-x")
        self.argv.append(str(Path(args.libraries_path) / args.organism_name / "index.Libs" / (str(args.organism_name) + "_genome")))
        self.tail = ["--phred64-quals"] if getattr(args, "phred64", False) else []

    def _run(self, opts: str, seqs: Sequence[str], fname: str) -> str:
        fa = self.workDir / fname
        with open(fa, "w") as fh:
            fh.write("".join(f">{s}\n{s}\n" for s in seqs))
        try:
            return subprocess.run(self.argv + shlex.split(opts) + [str(fa)] + self.tail, check=True, stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True).stdout
        finally:
            os.remove(fa)

    def unique_best(self, seqs: Sequence[str]) -> set:
        hits: Dict[str, List[int]] = {}
        for row in self._run(' -n 1 -f -a -3 2 ', seqs, "SeqToMap.fasta").split("\n"):
            if not row.startswith('@'):
                f = row.split('\t')
                if f != ['']:
                    hits.setdefault(f[0], []).append(f[-1].count(':'))
        return {s for s, c in hits.items() if len(c) == 1 or c.count(min(c)) == 1}

    def aligned(self, seqs: Sequence[str]) -> set:
        out = set()
        for row in self._run(' -n 0 -f -a -3 2 ', seqs, "SeqToJudge.fasta").split("\n"):
            if not row.startswith('@'):
                f = row.split('\t')
                if f != ['']:
                    out.add(f[0])
        return out


class ListedGenome:
    """The same two answers from lists computed elsewhere (one sequence per line, '#' comments)."""

    def __init__(self, retained: Iterable[str], aligned: Iterable[str] = ()):
        self._ret, self._al = set(retained), set(aligned)

    @staticmethod
    def from_files(retained_path, aligned_path=None):
        rd = lambda p: [ln.strip() for ln in open(p) if ln.strip() and not ln.startswith("#")]
        return ListedGenome(rd(retained_path), rd(aligned_path) if aligned_path else ())

    def unique_best(self, seqs):
        return {s for s in seqs if s in self._ret}

    def aligned(self, seqs):
        return {s for s in seqs if s in self._al}


# ---------------------------------------------------------------------------------------------------------------
# families and targets
# ---------------------------------------------------------------------------------------------------------------
def families(mirna_names: Sequence[str], merges: List[List[str]]):
    """-> (family names, fam_of_ref): the merged name of every miRNA reference (summary.py:707-712), else its own"""
    merged = {m: row[0] for row in merges for m in row[1:]}
    fam_index: Dict[str, int] = {}
    fam_of_ref = np.empty(len(mirna_names), dtype=np.int32)
    for r, nm in enumerate(mirna_names):
        fam_of_ref[r] = fam_index.setdefault(merged.get(nm, nm), len(fam_index))
    return list(fam_index), fam_of_ref


def read_pseudo_fasta(path) -> Dict[str, str]:
    """``<org>_mirna_SNP_pseudo_<db>.fa``: strictly two lines per record, as the reference reads it (:1037-1044)"""
    out = {}
    with open(path) as fh:
        lines = fh.read().split("\n")
    for k in range(0, len(lines) - 1, 2):
        if lines[k] == '':
            break
        out[lines[k].strip()[1:]] = lines[k + 1].strip()
    return out


def tally(casc: Cascade, uniq: _ffi.DeviceReads, res: _ffi.CascadeResult, fam_of_ref=None, targets: FlatSeqs = None,
          retained=None, freq=None, per_read: bool = False):
    """Counting core alone (bench.py's config 5): by default every miRNA is its own family with its own sequence as the
    canonical one, every read is retained and every isomiR read is a member."""
    mir = casc.libs["mirna"]
    if fam_of_ref is None:
        fam_of_ref, targets = np.arange(len(mir), dtype=np.int32), mir.seqs
    if freq is None:
        freq = np.full(uniq.n_samples, 1e300)
    return _ffi.variant_tally(casc.ctx, uniq, res, EXACT_PASS, ISO_PASS, fam_of_ref, targets, retained, freq, per_read)


def mismatch_census(census: np.ndarray, gate: np.ndarray) -> np.ndarray:
    """[S, 12, 3]: the twelve base changes (order of ``BASE_PAIRS``), raw / accepted / accepted-and-retained, summed
    over the (miRNA, sample) pairs that pass the report's gates -- ``mismatchCount.csv`` (:1148-1172, :1333-1348)"""
    S = census.shape[-1]
    out = np.zeros((S, 12, 3), dtype=np.int64)
    for k, (a, b) in enumerate(BASE_PAIRS):
        per = census[:, :, CODE[a], CODE[b], :, :].sum(axis=1)  # [fam, 3, S]
        out[:, k, :] = (per * gate[:, None, :]).sum(axis=0).T
    return out


# ---------------------------------------------------------------------------------------------------------------
# the report
# ---------------------------------------------------------------------------------------------------------------
def _refine_name(name: str) -> str:
    return name.replace('.fastq', '', 1) if '.fastq' in name else name  # refineName (:950-961): first occurrence


def a2i_report(args, workDir, ref_db, base_names, casc: Cascade, uniq, res, seqs: FlatSeqs, ps, ref, counts, order,
               tables: dict, merges, genome=None):
    """``a2i_editing`` (:979-1432).  ``tables`` = what ``finish_tables`` returned (miR.Counts frame, Filtered miRNA
    Reads); ``order`` = row order of the mapped frame; ``genome`` = predicate object (default: ``BowtieGenome``)."""
    from scipy import stats
    workDir = Path(workDir)
    base_names = list(base_names)
    S = len(base_names)
    mir = casc.libs["mirna"]
    fam_names, fam_of_ref = families(mir.names, merges)
    pseudo = read_pseudo_fasta(Path(args.libraries_path) / args.organism_name / "fasta.Libs" /
                               (args.organism_name + "_mirna_SNP_pseudo_" + ref_db + ".fa"))
    filtered = [tables["filtered"][b] for b in base_names]
    freq = np.array([(1000000 / f) if f != 0 else 0 for f in filtered], dtype=np.float64)
    # rows of the mapped frame that belong to the miRNA classes, in frame order: exact rows first, then isomiR rows
    # (canonical_gff then isomir_gff, :994-1016) -- the order of the families' lists and of the detail file
    is_ex, is_iso = ps[order] == EXACT_PASS, ps[order] == ISO_PASS
    idx = np.concatenate([order[is_ex], order[is_iso]])
    present = np.unique(fam_of_ref[ref[idx]]) if idx.size else np.zeros(0, np.int32)
    missing = [fam_names[f] for f in present if fam_names[f] not in pseudo]
    if missing:
        raise KeyError(f"{missing[:3]} ... have reads but no entry in the SNP-pseudo FASTA (the reference fails the same way, :1127)")
    targets = FlatSeqs.from_list([pseudo.get(n, "") for n in fam_names])
    genome = genome or BowtieGenome(args, workDir)
    # members (the kernel applies the same rule; needed here for the genome run's input and the detail file)
    cnt64 = counts.astype(np.int64)
    member = np.zeros(len(seqs), dtype=bool)
    member[order[is_ex]] = True
    iso_rows = order[is_iso]
    member[iso_rows] = ((cnt64[iso_rows].astype(np.float64) * freq[None, :]) >= 1).any(axis=1)
    midx = idx[member[idx]]
    mseqs = seqs.take(midx).to_list()
    retained_set = genome.unique_best(mseqs)
    retained = np.zeros(len(seqs), dtype=np.uint8)
    retained[midx] = [s in retained_set for s in mseqs]
    t = _ffi.variant_tally(casc.ctx, uniq, res, EXACT_PASS, ISO_PASS, fam_of_ref, targets, retained, freq, per_read=True)
    # ---- gates per (family, sample): the miRNA's RPM >= 1 and at least two sequences with a count (:1130-1137)
    mir_counts = tables["counts"]
    F = len(fam_names)
    gate = np.zeros((F, S), dtype=np.int64)
    for f in np.nonzero(t["n_seqs"].sum(axis=1) > 0)[0]:
        row = mir_counts.loc[fam_names[f]] if fam_names[f] in mir_counts.index else None
        for s in range(S):
            if row is None or filtered[s] == 0:
                continue
            if 1000000.0 * float(row[base_names[s]]) / filtered[s] >= 1 and t["n_seqs"][f, s] > 1:
                gate[f, s] = 1
    ag = t["census"][:, :, A, G, 2, :]  # [fam, 32, S]: A -> G, accepted and retained
    # ---- per (family, position, sample): count, ratio, p, BH
    cells: Dict[tuple, dict] = {}
    fam_order = list(dict.fromkeys(fam_of_ref[ref[midx]].tolist()))  # insertion order of mirNameSeqDicTmp
    for f in fam_order:
        for s in range(S):
            if not gate[f, s]:
                continue
            ct = int(t["count_true"][f, s])
            for q in np.nonzero(ag[f, :, s])[0]:
                c = int(ag[f, q, s])
                p = float(stats.binom.cdf(ct - c, ct, 1 - P_MISMATCH)) if ct - c >= 0 else 1.0
                cells[(f, int(q) + 1, s)] = dict(count_true=ct, canon=int(t["canon"][f, s]), count=c,
                                                 ratio=(c / ct if ct else 0), p=p, kept_exact=int(t["kept_exact"][f, s]))
    for s in range(S):  # Benjamini-Hochberg per sample (:1175-1183); ties on p are ordered by 'name:position'
        plist = sorted([v["p"], f"{fam_names[f]}:{q}", (f, q, s)] for (f, q, s2), v in cells.items() if s2 == s)
        for k, (p, _, key) in enumerate(plist):
            cells[key]["adj"] = p * len(plist) / (k + 1)
    # ---- tmp1 -> tmp2 (rows with a significant sample) -> tmp3 (sorted) -> final (RPM / repeat / genome filters)
    header = 'miRNA,A-to-I position in the miRNA,miRNA sequence'
    for b in base_names:
        n = _refine_name(b)
        header += ',' + ','.join([n + '.readCount', n + '.readCount.canonical', n + '.RPM.canonical', n + '.readCount.mismatch',
                                  n + '.RPM.mismatch', n + '.AtoI.percentage', n + '.AtoI.adjusted.pValue'])
    rows = []
    for (f, q) in dict.fromkeys((f, q) for (f, q, _s) in cells):
        fields, keep = [fam_names[f], str(q), pseudo[fam_names[f]]], False
        for s in range(S):
            v = cells.get((f, q, s))
            if v is None:
                fields += ['NE'] * 7
                continue
            part = [str(v["count_true"]), str(v["canon"]), '%.2f' % (1000000.0 * v["canon"] / filtered[s]), str(v["count"]),
                    '%.2f' % (1000000.0 * v["count"] / filtered[s])]
            if v["kept_exact"] > 0:
                part += ['%.2f%%' % (v["ratio"] * 100), ('%.2E' % v["adj"]) if v["adj"] <= 0.05 else 'NS']
                keep = keep or v["adj"] <= 0.05
            else:
                part += ['NE', 'NE']
            fields += part
        if keep:
            rows.append(fields)
    rows.sort(key=lambda r: (r[0].encode(), int(r[1])))  # sort -t',' -k1,1 -k2,2n under LC_ALL=C
    repeat_file = Path(args.libraries_path) / args.organism_name / 'annotation.Libs' / \
        (args.organism_name + '_miRNAs_in_repetitive_element_' + ref_db + ".csv")
    in_repeats = set()
    try:
        with open(repeat_file) as fh:
            in_repeats = {ln.strip().split(',')[0] for ln in fh}
    except FileNotFoundError:
        pass
    edited = [r[2][:int(r[1]) - 1] + 'G' + r[2][int(r[1]):] for r in rows]
    on_genome = genome.aligned(edited) if rows else set()
    final = []
    for r, e in zip(rows, edited):
        rpm = []
        for s in range(S):
            try:
                rpm.append(float(r[5 + s * 7]))
            except ValueError:
                rpm.append(0)
        if not any(x >= 1 for x in rpm) or r[0] in in_repeats or e in on_genome:
            continue
        final.append(r)
    with open(workDir / 'a2IEditing.report.csv', 'w') as fh:
        fh.write(header + '\n' + ''.join(','.join(r) + '\n' for r in final))
    # ---- the heat-map form (:1366-1416)
    labels = []
    for item in header.split(','):
        if '.AtoI.percentage' in item and item.split('.')[0] not in labels:
            labels.append(item.split('.')[0])
    kept, content = [], {}
    for r in final:
        per = []
        for k in range(0, len(r) - 3, 7):
            item = r[3 + k:3 + k + 7]
            per.append(('NA', 'NA') if item[6] in ('NE', 'NS') else (item[5][:-1], str(math.log(float(item[4]), 2))))
        ok = False
        for pct, _ in per:
            try:
                if float(pct) >= PCT_CUTOFF:
                    ok = True
                    break
            except ValueError:
                pass
        if ok:
            name = ":".join(r[:2])
            kept.append((sum(1 for pct, _ in per if pct != 'NA'), name))
            content[name] = per
    kept.sort(reverse=True)
    with open(workDir / 'a2IEditing.report.newform.csv', 'w') as fh:
        fh.write('miRNA:position,sample,A-to-I percentage,log2RPM\n')
        for k, lab in enumerate(labels):
            for _, name in kept:
                fh.write(name + ',' + SAMPLE_LABELS.get(lab, lab) + ',' + ','.join(content[name][k]) + '\n')
    # ---- mismatchCount.csv (the reference writes and removes it; kept here)
    mm = mismatch_census(t["census"], gate)
    with open(workDir / 'mismatchCount.csv', 'w') as fh:
        cols = ['>'.join(p) for p in BASE_PAIRS]
        fh.write('sample,' + ','.join([c + '_raw' for c in cols] + cols + [c + '_filtered' for c in cols]) + '\n')
        for s, b in enumerate(base_names):
            fh.write(b + ',' + ','.join(str(int(x)) for v in range(3) for x in mm[s, :, v]) + '\n')
    # ---- a2IEditing.detail.txt (:368-399): the aligned block of every (miRNA, sample) that passed the gates
    write_detail(workDir / 'a2IEditing.detail.txt', fam_order, fam_names, pseudo, gate, midx, fam_of_ref[ref[midx]], mseqs,
                 cnt64[midx], t["diag"][midx], t["state"][midx], retained[midx])
    return dict(report=final, tally=t, gate=gate, families=fam_names, retained=retained)


def write_detail(path, fam_order, fam_names, pseudo, gate, midx, fam_of, mseqs, mcounts, diag, state, retained):
    by_fam: Dict[int, List[int]] = {}
    for k, f in enumerate(fam_of.tolist()):
        by_fam.setdefault(f, []).append(k)
    S = gate.shape[1]
    out = []
    for f in fam_order:
        target = pseudo[fam_names[f]]
        for s in range(S):
            if not gate[f, s]:
                continue
            ks = [k for k in by_fam.get(f, []) if mcounts[k, s] > 0]
            ds = [int(diag[k]) for k in ks]
            H = max([0] + [-d for d in ds])
            T = max([0] + [d + len(mseqs[k]) - len(target) for k, d in zip(ks, ds)])
            width = H + len(target) + T
            lines = [("-" * (H + d) + mseqs[k] + "-" * (width - H - d - len(mseqs[k])), str(int(mcounts[k, s])), state[k] == 1,
                      bool(retained[k])) for k, d in zip(ks, ds)]
            out.append('Canonical_Seq of %s: %s\n' % (fam_names[f], target))
            out.append('seqList size is: %d, %d\n' % (len(ks), len(ks)))
            out += ['\t'.join([a, c, str(st)]) + '\n' for a, c, st, _ in lines]
            out.append('****************\n')
            out.append('retained seqList size is: %d\n' % sum(1 for _, _, st, rt in lines if st and rt))
            out += ['\t'.join([a, c, str(st)]) + '\n' for a, c, st, _ in lines if st]
            out.append('****************\n')
            out.append('retained sequences after filering are:\n')
            out += ['\t'.join([a, c, str(st)]) + '\n' for a, c, st, rt in lines if st and rt]
    with open(path, 'w') as fh:
        fh.write(''.join(out))
