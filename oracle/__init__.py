"""oracle -- CPU restatement of the miRge3.0 hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product path (``mirge3.0_amd``) never does.

PARITY STATUS -- "parity unpinned" for the alignment predicate: the cascade's arithmetic is
bowtie 1.x (third-party, absent here, no golden vectors in the reference); see the header of
``mirge_oracle.c``.  Pinned against the real reference code through ``tests/golden``:
pass order / subsets / overwrite rule (``mirge/libs/manifoldAlign.py:12-146``), the collapse
rule (``mirge/libs/digest.py:141-163``) and the whole count join (``mirge/libs/summary.py``).

* ``mirge_oracle.c``: cascade (brute force + indexed) and collapse, plain C.
* ``join`` below: the count join of ``summary.py:25-45,677-798,882-901,1223-1291`` in numpy.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmirge_oracle.so")
_SRC = os.path.join(_HERE, "mirge_oracle.c")


def build(force: bool = False) -> str:
    """gcc the C restatement into oracle/_build/ (git-ignored; travels to the GPU box)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", _SO, _SRC])
    return _SO


class _Policy(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "mode", "mm", "seedlen", "maxtotal", "trim5", "trim3", "ttail", "len_lt", "len_gt",
        "need_unannotated")]


class _Lib(C.Structure):
    _fields_ = [("data", C.c_void_p), ("offsets", C.c_void_p), ("n", C.c_int64)]


# The ten passes: column written, library key, bowtie argument string
# (mirge/libs/manifoldAlign.py:84-85) and the restated predicate.
PASSES = [
    # column,          lib key,        bowtie args (verbatim),                      policy
    ("exact miRNA",   "mirna",        "-n 0 -f --norc -S",
     dict(mode=0, mm=0, seedlen=28, maxtotal=2, len_lt=26)),
    ("hairpin miRNA", "hairpin",      "-n 1 -f --norc -S",
     dict(mode=0, mm=1, seedlen=28, maxtotal=2, len_gt=25)),
    ("mature tRNA",   "mature_trna",  "-v 1 -f -a --best --strata --norc -S",
     dict(mode=1, mm=1, seedlen=28, maxtotal=1, need_unannotated=1)),
    ("primary tRNA",  "pre_trna",     "-v 0 -f -a --best --strata --norc -S",
     dict(mode=1, mm=0, seedlen=28, maxtotal=0, ttail=1, need_unannotated=1)),
    ("snoRNA",        "snorna",       "-n 1 -f --norc -S",
     dict(mode=0, mm=1, seedlen=28, maxtotal=2, need_unannotated=1)),
    ("rRNA",          "rrna",         "-n 1 -f --norc -S",
     dict(mode=0, mm=1, seedlen=28, maxtotal=2, need_unannotated=1)),
    ("ncrna others",  "ncrna_others", "-n 1 -f --norc -S",
     dict(mode=0, mm=1, seedlen=28, maxtotal=2, need_unannotated=1)),
    ("mRNA",          "mrna",         "-n 0 -f --norc -S",
     dict(mode=0, mm=0, seedlen=28, maxtotal=2, need_unannotated=1)),
    ("isomiR miRNA",  "mirna",        "-5 1 -3 2 -v 2 -f --norc --best -S",
     dict(mode=1, mm=2, seedlen=28, maxtotal=2, trim5=1, trim3=2, need_unannotated=1)),
    ("spike-in",      "spike-in",     "-n 0 -f --norc -S",
     dict(mode=0, mm=0, seedlen=28, maxtotal=2, need_unannotated=1)),
]


def _lib():
    so = C.CDLL(build())
    so.oracle_cascade.restype = C.c_int
    so.oracle_collapse.restype = C.c_int64
    so.oracle_align_one.restype = C.c_int
    so.oracle_build_seconds.restype = C.c_double
    return so


def build_seconds(reset: bool = False) -> float:
    """Seconds the C cascade spent constructing its k-mer tables since the last reset."""
    return float(_lib().oracle_build_seconds(C.c_int(1 if reset else 0)))


def _policy(d: dict) -> _Policy:
    p = _Policy()
    for k, v in d.items():
        setattr(p, k, v)
    return p


def cascade(data: np.ndarray, offsets: np.ndarray,
            libs: Sequence[Optional[Tuple[np.ndarray, np.ndarray]]],
            n_pass: int = 9, indexed: bool = False, threads: int = 0,
            policies: Optional[Sequence[dict]] = None):
    """Run the cascade.  ``libs[p]`` = (ascii uint8, int64 offsets) of the library of pass p
    (or None to skip the pass).  Returns int32 arrays (pass, ref, off, mm), -1 = unannotated."""
    so = _lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.shape[0] - 1
    keep = []
    arr = (_Lib * n_pass)()
    for p in range(n_pass):
        lp = libs[p]
        if lp is None:
            arr[p].n = 0
            continue
        d = np.ascontiguousarray(lp[0], dtype=np.uint8)
        o = np.ascontiguousarray(lp[1], dtype=np.int64)
        keep += [d, o]
        arr[p].data, arr[p].offsets, arr[p].n = d.ctypes.data, o.ctypes.data, o.shape[0] - 1
    pol = (_Policy * n_pass)(*[_policy((policies[p] if policies else PASSES[p][3])) for p in range(n_pass)])
    out = [np.empty(max(n, 1), dtype=np.int32) for _ in range(4)]
    so.oracle_cascade(C.c_void_p(data.ctypes.data), C.c_void_p(offsets.ctypes.data), C.c_int64(n),
                      arr, pol, C.c_int32(n_pass), C.c_int32(1 if indexed else 0), C.c_int32(threads),
                      *[C.c_void_p(a.ctypes.data) for a in out])
    return tuple(a[:n] for a in out)


def collapse(data: np.ndarray, offsets: np.ndarray):
    """-> (first_index[U], count[U], inverse[n]); uniques in order of first appearance."""
    so = _lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.shape[0] - 1
    first = np.empty(max(n, 1), dtype=np.int64)
    cnt = np.empty(max(n, 1), dtype=np.int64)
    inv = np.empty(max(n, 1), dtype=np.int64)
    U = so.oracle_collapse(C.c_void_p(data.ctypes.data), C.c_void_p(offsets.ctypes.data), C.c_int64(n),
                           C.c_void_p(first.ctypes.data), C.c_void_p(cnt.ctypes.data),
                           C.c_void_p(inv.ctypes.data))
    return first[:U], cnt[:U], inv[:n]


# --------------------------------------------------------------------------------------
# count join (mirge/libs/summary.py)
# --------------------------------------------------------------------------------------
REPORT_CLASSES = [  # (report column, pass index)  summary.py:686-690,895-901
    ("Hairpin miRNAs", 1), ("mature tRNA Reads", 2), ("primary tRNA Reads", 3),
    ("snoRNA Reads", 4), ("rRNA Reads", 5), ("ncRNA others", 6), ("mRNA Reads", 7),
]


def umi_parser(s: str, f: int, b: int):
    """``UMIParser`` (digest.py:305-315): -> (insert, UMI bases).  ``s[-b:]`` with b == 0 is the whole
    read, so the 'UMI' of a front-only layout is front + read -- kept, it is what the reference writes."""
    center = s[f:-b] if int(b) != 0 else s[f:]
    return center, s[:f] + s[-b:]


def umi_collapse(raw: List[str], f: int, b: int, min_len: int, dedup: bool):
    """Collapse with ``-umi f,b`` [``-udd``] (digest.py:358-365 worker filter, :158-163 merge, :164-205 UMI
    stage) -> (ordered list of (insert, count), trimmed, rows of <sample>_umiCounts.csv or None)."""
    complete = {}
    for s in raw:
        if len(umi_parser(s, f, b)[0]) >= min_len:
            complete[s] = complete.get(s, 0) + 1
    d, trimmed, rows = {}, 0, []
    for s, c in complete.items():
        pure, tag = umi_parser(s, f, b)
        if len(pure) >= min_len:
            if dedup:
                rows.append(f"{tag},{pure},{c}\n")
            d[pure] = d.get(pure, 0) + (1 if dedup else c)
            trimmed += 1 if dedup else c
    return list(d.items()), trimmed, (["UMISeq,transcriptSeq,UMICounts\n"] + rows if dedup else None)


def qiagen_key(current: str, trimmed: str, adapter_len: int, b: int) -> str:
    """The dictionary key of the worker's ``qiagenumi`` branch (digest.py:340-348): the trimmed read plus the UMI that
    follows the 3' adapter in the untrimmed read, by the reference's own string rule --
    ``currentSeq.split(trimmed)[1][:len(adapter)+b][-b:]`` (first occurrence of the trimmed read; the text up to its
    next occurrence; ``ValueError`` of an empty separator -> no UMI).  [The reference would raise IndexError if the
    trimmed read did not occur in the untrimmed one; a modifier chain that only removes bases cannot produce that.]"""
    try:
        umi_seq = current.split(trimmed)[1]
        max_ad = adapter_len + int(b)
        umi_seq = umi_seq[:max_ad][-int(b):]
    except ValueError:
        umi_seq = ""
    return trimmed + umi_seq


def umi_worker_reads(records, opts: dict, f: int, b: int, min_len: int, qiagen: bool, per_modifier: bool = True):
    """What the per-chunk worker puts into its dictionary with ``-umi`` (digest.py:334-365), one entry per count, in
    file order.  records: (sequence, qualities | None).  qiagen: the read after the LAST modifier + its UMI, kept when
    the trimmed read has min_len bases (:349).  Otherwise: the read after EVERY modifier (``per_modifier``, the loop
    body at HEAD :354-365) or after the last, kept when ``UMIParser`` leaves min_len bases (:359-360)."""
    out = []
    for seq, qual in records:
        stages = trim_stages(seq, qual, opts)
        if qiagen:
            final = stages[-1] if stages else seq
            if len(final) >= min_len:
                out.append(qiagen_key(seq, final, len(opts["adapter"]), b))
            continue
        for st in (stages if per_modifier else stages[-1:]):
            if len(umi_parser(st, f, b)[0]) >= min_len:
                out.append(st)
    return out


def umi_baking(keys: List[str], f: int, b: int, min_len: int, dedup: bool):
    """baking's merge (digest.py:158-163) and UMI stage (:164-205) over the worker's keys ->
    (ordered (insert, count) pairs, trimmed, lines of <sample>_umiCounts.csv or None)."""
    complete = {}
    for s in keys:
        complete[s] = complete.get(s, 0) + 1
    d, trimmed, rows = {}, 0, []
    for s, c in complete.items():
        pure, tag = umi_parser(s, f, b)
        if len(pure) >= min_len:
            if dedup:
                rows.append(f"{tag},{pure},{c}\n")
            d[pure] = d.get(pure, 0) + (1 if dedup else c)
            trimmed += 1 if dedup else c
    return list(d.items()), trimmed, (["UMISeq,transcriptSeq,UMICounts\n"] + rows if dedup else None)


def _fmt_float(x: float) -> str:
    return repr(float(x))


def join(pass_: np.ndarray, ref: np.ndarray, counts: np.ndarray, mirna_names: List[str],
         mirna_headers: List[str], merges: List[List[str]], samples: List[str],
         sample_read_counts: Dict[str, int], trimmed_read_counts: Dict[str, int],
         trimmed_unique: Dict[str, int], cr_threshold: float = 0.1, spike: bool = False):
    """Restates summary.py: per-class sums (:686-698), exact/isomiR group-by (:749-752),
    mirge_can (:25-45), merged-family renaming (:759-764), totals (:766-771,882-888), the
    full name universe (:774-790) and the three CSV texts (:796-797,1281-1284)."""
    S = len(samples)
    counts = np.asarray(counts, dtype=np.int64).reshape(-1, S)
    R = len(mirna_names)
    class_sums = {}
    classes = list(REPORT_CLASSES) + ([("Spike-in", 9)] if spike else [])
    for col, p in classes:
        class_sums[col] = counts[pass_ == p].sum(axis=0)
    exact = np.zeros((R, S), dtype=np.int64)
    iso = np.zeros((R, S), dtype=np.int64)
    m0, m8 = pass_ == 0, pass_ == 8
    np.add.at(exact, ref[m0], counts[m0])
    np.add.at(iso, ref[m8], counts[m8])
    all_mirna = counts[m0 | m8].sum(axis=0)  # 'All miRNA Reads' summary.py:769-771
    has_exact = np.zeros(R, dtype=bool)
    has_exact[ref[m0]] = True  # groupby keeps names that occur, even with zero counts
    merged_of = {}
    universe: Dict[str, int] = {}
    for row in merges:
        for item in row[1:]:
            merged_of[item] = row[0]
            universe[row[0]] = 1
    # mirge_can per sample, over the names that have exact rows (left merge)
    filt: Dict[str, np.ndarray] = {}
    # pandas dtype flow of mirge_can: a sample column stays int64 only if every exact-name row
    # passes the ratio test (no NaN is introduced at summary.py:41); one NaN makes it float64
    col_all_pass = np.ones(S, dtype=bool)
    for r in np.nonzero(has_exact)[0]:
        x = exact[r].astype(np.float64)
        y = iso[r].astype(np.float64)
        low = x < 2
        x = np.where(low, 0.0, x)
        y = np.where(low, 0.0, y)
        ratio = np.where(y > 0, x / np.where(y > 0, y, 1.0), x)
        ok = ratio > cr_threshold
        col_all_pass &= ok
        val = np.where(ok, x + y, 0.0)
        name = merged_of.get(mirna_names[r], mirna_names[r])
        filt[name] = filt.get(name, 0.0) + val
    names_sorted = sorted(filt)
    table = np.array([filt[nm] for nm in names_sorted], dtype=np.float64).reshape(-1, S)
    filtered = table.sum(axis=0) if len(names_sorted) else np.zeros(S)
    with np.errstate(divide="ignore", invalid="ignore"):
        rpm = np.round(table / table.sum(axis=0) * 1e6, 4) if len(names_sorted) else table
    unique_mirnas = (table > 0).sum(axis=0) if len(names_sorted) else np.zeros(S, dtype=np.int64)
    for h in mirna_headers:  # bowtie-inspect -n lines, summary.py:783-788
        srow = h.split(" ")[0] if "segs:" in h else h
        if srow not in merged_of:
            universe[srow] = 1
    all_names = sorted(set(universe) | set(names_sorted))
    idx = {nm: i for i, nm in enumerate(names_sorted)}
    outer_adds_nan = any(nm not in idx for nm in all_names)  # join(how='outer').fillna(0), :792

    def table_csv(tab: np.ndarray, int_cols) -> str:
        lines = ["miRNA," + ",".join(samples)]
        for nm in all_names:
            if nm in idx:
                vals = [0.0 if np.isnan(v) else v for v in tab[idx[nm]]]
            else:
                vals = [0.0] * S
            lines.append(nm + "," + ",".join(str(int(v)) if int_cols[j] else _fmt_float(v)
                                             for j, v in enumerate(vals)))
        return "\n".join(lines) + "\n"

    counts_int = [bool(col_all_pass[j]) and not outer_adds_nan for j in range(S)]
    cols = ["Total Input Reads", "Trimmed Reads (all)", "Trimmed Reads (unique)", "All miRNA Reads",
            "Filtered miRNA Reads", "Unique miRNAs"] + [c for c, _ in classes] + ["Remaining Reads"]
    rep_lines = ["Sample name(s)," + ",".join(cols)]
    report = {}
    for s, nm in enumerate(samples):
        row = {
            "Total Input Reads": int(sample_read_counts[nm]),
            "Trimmed Reads (all)": int(trimmed_read_counts[nm]),
            "Trimmed Reads (unique)": int(trimmed_unique[nm]),
            "All miRNA Reads": int(all_mirna[s]),
            "Filtered miRNA Reads": int(filtered[s]),
            "Unique miRNAs": int(unique_mirnas[s]),
        }
        for c, _ in classes:
            row[c] = int(class_sums[c][s])
        row["Remaining Reads"] = row["Trimmed Reads (all)"] - (row["All miRNA Reads"] + sum(row[c] for c, _ in classes))
        report[nm] = row
        rep_lines.append(nm + "," + ",".join(str(row[c]) for c in cols))
    return dict(class_sums=class_sums, exact=exact, iso=iso, names=names_sorted, table=table, rpm=rpm,
                report=report, counts_csv=table_csv(table, counts_int), rpm_csv=table_csv(rpm, [False] * S),
                report_csv="\n".join(rep_lines) + "\n")


# --------------------------------------------------------------------------------------
# A-to-I editing (row a16 / N1), restated on strings from mirge/libs/mirge2_tRF_a2i.py:230-518.  PINNED against
# the reference's own align2TargetSeq / judgeAllign / A2IEditing / mismatchCountAnalysis through
# tests/golden/case4_gff_a2i/a2i_direct.json (made by tests/golden/make_golden.py with a Bio.pairwise2 stand-in:
# Biopython is absent, so the ALIGNER is restated from its documented behaviour -- best local alignment, +2 / -1,
# a gap costs 20 per character so it never pays for reads the cascade annotated; ties between diagonals are not
# pinned and the fixtures hold none).
# --------------------------------------------------------------------------------------
def local_diagonal(target: str, read: str):
    """Best ungapped local alignment of ``read`` on ``target`` (pairwise2.align.localms(target, read, 2, -1, -20, -20)
    for every pair whose optimum has no gap, mirge2_tRF_a2i.py:254): -> (d, score) with d = position of read base 0
    relative to target base 0.  Ties: the alignment that ENDS first in the target, then first in the read."""
    best = (0, None, None)  # (score, end_i, end_j)
    best_d = None
    lt, lr = len(target), len(read)
    for d in range(-(lr - 1), lt):
        run = 0
        i0 = max(d, 0)
        for i in range(i0, min(lt, d + lr)):
            j = i - d
            run = max(0, run + (2 if target[i] == read[j] else -1))
            if run > 0:
                key = (run, -(i + 1), -(j + 1))
                if best[1] is None or key > (best[0], -best[1], -best[2]):
                    best, best_d = (run, i + 1, j + 1), d
    return best_d, best[0]


def padded_pair(target: str, read: str, d: int):
    """the two strings as pairwise2 returns them for diagonal d: full sequences, '-' padded to one length"""
    hd_t, hd_s = max(0, -d), max(0, d)
    a, b = "-" * hd_t + target, "-" * hd_s + read
    n = max(len(a), len(b))
    return a + "-" * (n - len(a)), b + "-" * (n - len(b))


def judge_align(tgt: str, sq: str) -> bool:
    """judgeAllign (mirge2_tRF_a2i.py:298-332) on the padded pair"""
    def dashes(s):
        h = len(s) - len(s.lstrip("-"))
        t = len(s) - len(s.rstrip("-"))
        return h, t
    hd_t, td_t = dashes(tgt)
    hd_s, td_s = dashes(sq)
    len1 = len(tgt) - hd_t - td_t
    match_limit = len1 - 3 - 1
    end_pos1 = len(tgt) - hd_t - 1 - 3
    end_pos2 = len(sq) - td_s - 1
    if hd_s - hd_t > 1:
        return False
    mism = match = 0
    for pos in range(hd_t, min(end_pos1, end_pos2) + 1):
        if sq[pos] == "-":
            continue
        if tgt[pos] != sq[pos]:
            mism += 1
        else:
            match += 1
    if hd_s - hd_t == 1:
        match_limit -= 1
    return not (mism > 1 or match < match_limit)


BASE_PAIRS = [('A', 'G'), ('A', 'C'), ('A', 'T'), ('T', 'G'), ('T', 'A'), ('T', 'C'), ('C', 'G'), ('C', 'A'), ('C', 'T'),
              ('G', 'A'), ('G', 'C'), ('G', 'T')]  # the order of mismatchCountAnalysis (:440)


def a2i_group(target: str, reads: List[str], counts: List[int], retained, start_base="A", end_base="G"):
    """A2IEditing (mirge2_tRF_a2i.py:335-419) for one miRNA and one sample -> dict with the reference's return values
    (positions in order of first appearance) plus the per-read diagonal / state and the aligned block of the detail
    file; mismatchCountAnalysis (:424-518) -> ``census`` [12][3] (raw, state-true, state-true & retained)."""
    from scipy import stats
    ds = [local_diagonal(target, r)[0] for r in reads]
    states = [judge_align(*padded_pair(target, r, d)) for r, d in zip(reads, ds)]
    H = max([0] + [-d for d in ds])                       # head dashes of the target in the joint frame
    T = max([0] + [d + len(r) - len(target) for r, d in zip(reads, ds)])
    width = H + len(target) + T
    frame = ["-" * H + target + "-" * T] + ["-" * (H + d) + r + "-" * (width - H - d - len(r)) for r, d in zip(reads, ds)]
    lt = len(target)
    positions, pcount, kept = [], {}, []
    count_true = seq_true = canon = 0
    for j, r in enumerate(reads):
        if states[j] and r in retained:
            if r in target:
                canon += counts[j]
            kept.append(frame[j + 1])
            seq_true += 1
            count_true += counts[j]
            for q in range(0, lt - 5):
                if target[q] == start_base and frame[j + 1][H + q] == end_base:
                    if q + 1 not in pcount:
                        positions.append(q + 1)
                        pcount[q + 1] = 0
                    pcount[q + 1] += counts[j]
    ratio = {p: (pcount[p] / count_true if count_true else 0) for p in positions}
    pval = {p: (float(stats.binom.cdf(count_true - pcount[p], count_true, 1 - 0.001)) if count_true - pcount[p] >= 0 else 1.0)
            for p in positions}
    census = []
    for a, b in BASE_PAIRS:
        tot = [0, 0, 0]
        for j, r in enumerate(reads):
            for q in range(0, lt - 5):
                if target[q] == a and frame[j + 1][H + q] == b:
                    tot[0] += counts[j]
                    if states[j]:
                        tot[1] += counts[j]
                        if r in retained:
                            tot[2] += counts[j]
        census.append(tot)
    return dict(diagonals=ds, states=states, frame=frame, kept=kept, positions=positions, count=pcount, ratio=ratio,
                pvalue=pval, countSumTrue=count_true, seqCountTrue=seq_true, canonicalSeqCount=canon, census=census)


def variant_tally(seqs: List[str], counts: np.ndarray, pass_: np.ndarray, ref: np.ndarray, fam_of_ref: np.ndarray,
                  targets: List[str], retained=None, freq=None, exact_pass: int = 0, iso_pass: int = 8, maxpos: int = 32):
    """What ``mirge_variant_tally`` returns, from the string restatements above: membership (:988-1016), alignment,
    judgeAllign, and the per-(family, sample) counts / per-position base-change census in three variants."""
    S = counts.shape[1]
    F = len(targets)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    out = {k: np.zeros((F, S), dtype=np.int64) for k in ("n_seqs", "seq_true", "count_true", "canon", "kept_exact")}
    census = np.zeros((F, maxpos, 4, 4, 3, S), dtype=np.int64)
    diag = np.zeros(len(seqs), dtype=np.int8)
    state = np.full(len(seqs), -1, dtype=np.int8)
    for i, read in enumerate(seqs):
        p = int(pass_[i])
        if p != exact_pass and p != iso_pass:
            continue
        f = int(fam_of_ref[int(ref[i])])
        if f < 0:
            continue
        if p == iso_pass and freq is not None and not any(float(counts[i, s]) * float(freq[s]) >= 1 for s in range(S)):
            continue
        target = targets[f]
        d, _ = local_diagonal(target, read)
        if d is None:
            d = 0
        st = judge_align(*padded_pair(target, read, d))
        diag[i], state[i] = d, 1 if st else 0
        keep = st and (retained is None or bool(retained[i]))
        for s in range(S):
            c = int(counts[i, s])
            if not c:
                continue
            out["n_seqs"][f, s] += 1
            if keep:
                out["seq_true"][f, s] += 1
                out["count_true"][f, s] += c
                if read in target:
                    out["canon"][f, s] += c
                if p == exact_pass:
                    out["kept_exact"][f, s] += 1
            for q in range(max(0, d), len(target) - 5):
                rj = q - d
                if rj >= len(read):
                    break
                b = read[rj]
                if b not in code or b == target[q]:
                    continue
                census[f, q, code[target[q]], code[b], 0, s] += c
                if st:
                    census[f, q, code[target[q]], code[b], 1, s] += c
                if keep:
                    census[f, q, code[target[q]], code[b], 2, s] += c
    out["census"], out["diag"], out["state"] = census, diag, state
    return out


# --------------------------------------------------------------------------------------
# isomiR typing for the miRTop GFF3 (row N2), restated from create_gff (mirge/libs/summary.py:48-606).  The reference
# diffs the canonical sequence against the read with difflib.Differ -- stdlib, so the oracle calls it too -- and then
# rewrites the two aligned lists in place; those rewrites are restated here on explicit index loops that follow
# Python's list-iterator semantics (an element deleted during iteration shifts the rest under the cursor).
# PINNED against the file the reference's create_gff wrote (tests/golden/case4_gff_a2i/sample_miRge3.gff).
# --------------------------------------------------------------------------------------
def _aligned_lists(master: str, read: str):
    from difflib import Differ
    result = list(Differ().compare(master, read))
    m = list(master)
    k = 0
    while k < len(m):  # summary.py:229-233: a '-' goes into the canonical list wherever the diff inserts a base
        if result[k].startswith("+"):
            m.insert(k, "-")
        k += 1
    sub = ["_" if r.startswith("-") else r.replace(" ", "") for r in result]
    m += ["-"] * (len(sub) - len(m))
    return m, sub


def _merge_replacements(m: list, sub: list):
    """the two in-place passes of summary.py:249-289 (a deletion next to an insertion is one substitution)"""
    y = 0
    while y < len(m):  # forward: '_' just before a '-'
        try:
            if y > 0 and m[y] == "-" and sub[y - 1] == "_":
                if y - 2 > 0 and sub[y - 2] == "_" and m[y + 1] == "-":
                    del m[y:y + 2]
                    del sub[y - 2:y]
                else:
                    m.pop(y)
                    sub.pop(y - 1)
        except IndexError:
            pass
        y += 1
    y = 0
    while y < len(m):  # reverse: '_' just after a '-'
        try:
            if y > 0 and m[y] == "-" and sub[y + 1] == "_":
                if y + 2 <= len(sub) and sub[y + 2] == "_" and m[y + 1] == "-":
                    del m[y:y + 2]
                    del sub[y:y + 2]
                else:
                    m.pop(y)
                    sub.pop(y + 1)
        except IndexError:
            pass
        y += 1


SNV_CLASSES = ["iso_snv_central_offset", "iso_snv_seed", "iso_snv_central", "iso_snv_central_supp", "iso_snv"]


def gff_record(master: str, read: str, precursor: str):
    """-> (type, start, end, variant, cigar) of one read against its miRNA (summary.py:170-470)"""
    if precursor != "":
        start = precursor.find(master) + 1
    else:
        start = 1
    end = start + len(master) - 1
    if read == master:
        return "ref_miRNA", start, end, "NA", str(len(read)) + "M"
    m, sub = _aligned_lists(master, read)
    _merge_replacements(m, sub)
    add, dele, subst = {}, {}, {}
    for k, v in enumerate(m):
        if v == "-":
            add[k] = sub[k]
        elif sub[k] == "_":
            dele[k] = v
        elif v != sub[k]:
            subst[k] = sub[k]
    a5 = d5 = a3 = d3 = ""
    for k in range(len(m)):
        if k in add:
            a5 += add[k]
        elif k in dele:
            d5 += dele[k]
        else:
            break
    for k in range(len(m), -1, -1):
        if k - 1 in add:
            a3 += add[k - 1]
        elif k - 1 in dele:
            d3 += dele[k - 1]
        else:
            break
    variant = ""
    if a5:
        s5 = a5.replace("+", "")
        ctx = list(precursor[start - len(s5) - 1:start - 1])
        try:
            t = sum(1 for k, c in enumerate(s5) if c == ctx[k])
            nt = len(s5) - t
            if t:
                variant += "iso_5p:+" + str(t) + ","
            if nt:
                variant += "iso_add5p:+" + str(nt) + ","
        except IndexError:
            variant += "iso_5p:-" + str(len(s5)) + ","
        start -= len(s5)
    if d5:
        variant += "iso_5p:+" + str(len(d5)) + ","
        start += len(d5)
    if a3:
        s3 = a3[::-1].replace("+", "")
        ctx = list(precursor[end:end + len(s3)])
        try:
            t = sum(1 for k, c in enumerate(s3) if c == ctx[k])
            nt = len(s3) - t
            if t:
                variant += "iso_3p:+" + str(t) + ","
            if nt:
                variant += "iso_add3p:+" + str(nt) + ","
        except IndexError:
            variant += "iso_3p:+" + str(len(s3)) + ","
        end += len(s3)
    if d3:
        variant += "iso_3p:-" + str(len(d3)) + ","
        end -= len(d3)
    seen = []
    for k in subst:
        cls = SNV_CLASSES[0] if k == 7 else SNV_CLASSES[1] if 1 <= k <= 6 else SNV_CLASSES[2] if 8 <= k <= 12 else \
            SNV_CLASSES[3] if 13 <= k <= 17 else SNV_CLASSES[4]
        if cls not in seen:
            seen.append(cls)
    variant += "".join(c + "," for c in seen)
    if variant.endswith(","):
        variant = variant[:-1]
    # CIGAR (summary.py:427-463): M for a column the two lists agree on, for an insertion and for a deletion; the
    # canonical base for a substitution; run-length encoded, a run of one printed without its count
    case = ""
    for k, v in enumerate(m):
        case += "M" if (v == sub[k] or v == "-" or sub[k] == "_") else v
    case = case.replace("+", "")
    cigar, run = "", 0
    last = ""
    for k, ch in enumerate(case):
        if k != 0:
            if ch == case[k - 1]:
                run += 1
            else:
                cigar += (str(run) if run != 1 else "") + case[k - 1]
                run = 1
        else:
            run += 1
        last = ch
    cigar += (str(run) if run != 1 else "") + last
    if not any(c in case for c in "ATGC"):
        cigar = str(len(read)) + "M"
    return "isomiR", start, end, variant or "iso_snv", cigar


# --------------------------------------------------------------------------------------
# read trimming (row N4): the modifier chain the reference builds from cutadapt (mirge/libs/digest.py:59-101) and the
# way its worker counts reads (:320-375).  PARITY UNPINNED: cutadapt is third-party and absent here, and the reference
# holds no vectors for it -- these functions restate cutadapt 2.x-4.x's published algorithms (qualtrim.pyx:
# quality_trim_index / nextseq_trim_index; _align.pyx: Aligner.locate for a 3' adapter; modifiers.py) from its
# documentation and sources as remembered; full-matrix Python, the checker of k_trim.
# --------------------------------------------------------------------------------------
def quality_trim_index(qual: str, cutoff_front: int, cutoff_back: int, base: int = 33):
    s = mx = start = 0
    for i in range(len(qual)):
        s += cutoff_front - (ord(qual[i]) - base)
        if s < 0:
            break
        if s > mx:
            mx, start = s, i + 1
    stop, s, mx = len(qual), 0, 0
    for i in reversed(range(len(qual))):
        s += cutoff_back - (ord(qual[i]) - base)
        if s < 0:
            break
        if s > mx:
            mx, stop = s, i
    if start >= stop:
        start, stop = 0, 0
    return start, stop


def nextseq_trim_index(seq: str, qual: str, cutoff: int, base: int = 33):
    s = mx = 0
    stop = len(qual)
    for i in reversed(range(len(qual))):
        q = ord(qual[i]) - base
        if seq[i] == "G":
            q = cutoff - 1
        s += cutoff - q
        if s < 0:
            break
        if s > mx:
            mx, stop = s, i
    return stop


def adapter_locate_back(adapter: str, read: str, max_error_rate: float = 0.12, min_overlap: int = 3, indels: bool = True,
                        read_wildcards: bool = False, adapter_wildcards: bool = True, anchored: bool = False):
    """Aligner.locate for a regular 3' adapter (the alignment may start anywhere in the read, stop anywhere in it, and
    stop inside the adapter when it runs off the read's end): unit costs, indels allowed; of the alignments with
    cost <= aligned adapter length * max_error_rate and at least min_overlap adapter bases, the one with the most
    matches, then the lowest cost, first found (full-adapter matches in order of their end in the read, then the
    partial ones at the read's end from the longest adapter prefix down).  An 'N' in the adapter matches any base and does not
    count towards the length the error rate applies to.  -> (astart, astop, rstart, rstop, matches, errors) or None.
    ``anchored`` (`-a ADAPTER$`, cutadapt's SuffixAdapter, flag START_WITHIN_SEQ2 alone): the alignment may start anywhere in the
    read but must take the WHOLE adapter and end at the read's last base -- one candidate, cell (m, n)."""
    m, n = len(adapter), len(read)
    wild = [c == "N" and adapter_wildcards for c in adapter]
    nwild = [0] * (m + 1)
    for i in range(m):
        nwild[i + 1] = nwild[i] + (1 if wild[i] else 0)
    # --no-indels: cutadapt keeps the same matrix and prices an insertion / a deletion at 100 000: neither is ever taken, and
    # an alignment cannot skip adapter bases in front of the read either (first column)
    INF = 10 ** 6
    prev = [(i if indels or i == 0 else INF, 0, 0) for i in range(m + 1)]  # (cost, matches, origin)
    best = None

    def consider(entry, i, j):
        nonlocal best
        cost, matches, origin = entry
        length = i
        eff = length - nwild[i]
        if length >= min_overlap and cost <= eff * max_error_rate and \
                (best is None or matches > best[4] or (matches == best[4] and cost < best[5])):
            best = (0, i, origin, j, matches, cost)

    done = False
    for j in range(1, n + 1):
        cur = [(0, 0, j)] + [None] * m
        for i in range(1, m + 1):
            d, up, left = prev[i - 1], cur[i - 1], prev[i]
            if wild[i - 1] or adapter[i - 1] == read[j - 1] or (read_wildcards and read[j - 1] == "N"):
                cur[i] = (d[0], d[1] + 1, d[2])
            else:
                cd, cdel, cins = d[0] + 1, (left[0] + 1 if indels else INF), (up[0] + 1 if indels else INF)
                if cd <= cdel and cd <= cins:
                    cur[i] = (cd, d[1], d[2])
                elif cins <= cdel:
                    cur[i] = (cins, up[1], up[2])
                else:
                    cur[i] = (cdel, left[1], left[2])
        prev = cur
        if not anchored or j == n:
            consider(cur[m], m, j)
        if best is not None and best[5] == 0 and best[4] == m and best[1] == m:
            done = True
            break
    if anchored:
        return best
    if not done:
        # cutadapt walks the last column from the longest adapter prefix down (`for i in reversed(range(first_i, m + 1))`):
        # of two prefixes with equal (matches, cost) the LONGER one is kept
        for i in range(m, -1, -1):
            consider(prev[i], i, n)
    return best


def adapter_locate_front(adapter: str, read: str, max_error_rate: float = 0.12, min_overlap: int = 3, indels: bool = True,
                         read_wildcards: bool = False, adapter_wildcards: bool = True, anchored: bool = False):
    """Aligner.locate for a regular 5' adapter (flags START_WITHIN_SEQ1 | START_WITHIN_SEQ2 | STOP_WITHIN_SEQ2: the
    alignment may start anywhere in the read AND inside the adapter -- row i of the first column costs 0 and has origin
    -i -- but must reach the adapter's last base): every read column is a candidate end; aligned adapter length =
    m + min(origin, 0); same acceptance (cost <= length * max_error_rate, length >= min_overlap) and the same order of
    preference as the 3' form (most matches, then lowest cost, first found).  No N in the adapter here.
    -> (astart, astop, rstart, rstop, matches, errors) or None; the read keeps read[rstop:].
    ``anchored`` (`-g ^ADAPTER`, and the 5' part of `-a ADAPTER1...ADAPTER2`; cutadapt's PrefixAdapter, flag STOP_WITHIN_SEQ2
    alone): both sequences start at their first base -- first row and first column cost their index -- and the whole adapter
    must be taken (aligned length m); the candidates are still the last row's cells, column by column."""
    m, n = len(adapter), len(read)
    assert "N" not in adapter.upper() or not adapter_wildcards
    INF = 10 ** 6
    prev = [((i if indels or i == 0 else INF), 0, 0) if anchored else (0, 0, -i) for i in range(m + 1)]  # (cost, matches, origin)
    best = None
    for j in range(1, n + 1):
        cur = [((j if indels else INF), 0, 0) if anchored else (0, 0, j)] + [None] * m
        for i in range(1, m + 1):
            d, up, left = prev[i - 1], cur[i - 1], prev[i]
            if adapter[i - 1] == read[j - 1] or (read_wildcards and read[j - 1] == "N"):
                cur[i] = (d[0], d[1] + 1, d[2])
            else:
                cd, cdel, cins = d[0] + 1, (left[0] + 1 if indels else 10 ** 6), (up[0] + 1 if indels else 10 ** 6)
                if cd <= cdel and cd <= cins:
                    cur[i] = (cd, d[1], d[2])
                elif cins <= cdel:
                    cur[i] = (cins, up[1], up[2])
                else:
                    cur[i] = (cdel, left[1], left[2])
        prev = cur
        cost, matches, origin = cur[m]
        length = m + min(origin, 0)
        if length >= min_overlap and cost <= length * max_error_rate and \
                (best is None or matches > best[4] or (matches == best[4] and cost < best[5])):
            best = (max(-origin, 0), m, max(origin, 0), j, matches, cost)
            if cost == 0 and matches == m:
                break
    return best


def trim_stages(seq: str, qual, opts: dict):
    """The read after each modifier of the chain (digest.py:59-101 builds it in this order): NextSeq quality trimming,
    quality trimming, adapter removal (a 3' adapter, or with opts["front"] a 5' adapter), N trimming at both ends,
    unconditional cuts."""
    out = []
    if opts.get("nextseq") is not None and qual is not None:
        stop = nextseq_trim_index(seq, qual, opts["nextseq"], opts.get("base", 33))
        seq, qual = seq[:stop], qual[:stop]
        out.append(seq)
    if opts.get("q_back") is not None and qual is not None:
        a, b = quality_trim_index(qual, opts.get("q_front", 0), opts["q_back"], opts.get("base", 33))
        seq, qual = seq[a:b], qual[a:b]
        out.append(seq)
    indels = opts.get("indels", True)
    times = int(opts.get("times", 1))
    rw, aw = bool(opts.get("read_wildcards", False)), bool(opts.get("adapter_wildcards", True))
    if opts.get("action") == "none" and (opts.get("adapter") or opts.get("adapters") or opts.get("linked")):
        out.append(seq)  # searched, not removed: the AdapterCutter still is a modifier of the chain
        opts = {k: v for k, v in opts.items() if k not in ("adapter", "adapters", "linked")}
    if opts.get("adapter") and not opts.get("adapters") and (times > 1 or not indels or rw or not aw or opts.get("anchored")):
        opts = dict(opts, adapters=[("front" if opts.get("front") else "back", opts["adapter"], bool(opts.get("anchored")))])
    if opts.get("linked"):
        # ONE linked adapter (cutadapt's LinkedAdapter.match_to): the 5' part is searched first -- anchored or regular --, and
        # when it is required and absent there is no match; the 3' part is searched in what follows the 5' match (the whole
        # read when an optional 5' part was not found); no 3' match is still a match when the 3' part is optional AND the 5'
        # part was found.  `-a A...B`: A anchored and required, B optional.  `-g A...B`: A regular, both required.  A match
        # removes whichever parts were found.  Restated from cutadapt's parser.py / adapters.py as remembered: parity unpinned.
        lk = opts["linked"]
        for _ in range(times):
            up = seq.upper()
            fm = adapter_locate_front(lk["front"], up, opts.get("error_rate", 0.12), opts.get("overlap", 3), indels, rw, aw,
                                      anchored=lk.get("front_anchored", False))
            if fm is None and lk.get("front_required", True):
                break
            lo = fm[3] if fm is not None else 0
            bm = adapter_locate_back(lk["back"], up[lo:], opts.get("error_rate", 0.12), opts.get("overlap", 3), indels, rw, aw,
                                     anchored=lk.get("back_anchored", False))
            if bm is None and (lk.get("back_required", False) or fm is None):
                break
            hi = lo + bm[2] if bm is not None else len(seq)
            seq = seq[lo:hi]
            qual = qual[lo:hi] if qual is not None else None
        out.append(seq)
    elif opts.get("adapters"):
        # AdapterCutter over several adapters, times = 1 (cutadapt's `_best_match`): every adapter is searched in the read
        # as it stands, the match with the most matching bases wins, then the one with fewer errors, then the first in the
        # list; only that ONE adapter is removed.  Restated from cutadapt's sources as remembered: parity unpinned.
        for _ in range(times):  # -n COUNT: `for _ in range(self.times): match = best_match(...); if match is None: break`
            best = None
            for spec in opts["adapters"]:
                kind, ad = spec[0], spec[1]
                anch = len(spec) > 2 and bool(spec[2])  # (kind, sequence, anchored): `-g ^ADAPTER` / `-a ADAPTER$`
                # (cutadapt's aligner translates lower-case letters like upper-case ones -- `_acgt_table`: "Lowercase versions are
                # also translated" --: the search is case-blind for either kind of adapter, the read keeps its letters)
                hit = (adapter_locate_front if kind == "front" else adapter_locate_back)(
                    ad, seq.upper(), opts.get("error_rate", 0.12), opts.get("overlap", 3), indels, rw, aw, anchored=anch)
                if hit is not None and (best is None or hit[4] > best[1][4] or (hit[4] == best[1][4] and hit[5] < best[1][5])):
                    best = (kind, hit)
            if best is None:
                break
            lo, hi = (best[1][3], len(seq)) if best[0] == "front" else (0, best[1][2])
            seq = seq[lo:hi]
            qual = qual[lo:hi] if qual is not None else None
        out.append(seq)
    elif opts.get("adapter") and opts.get("front"):
        hit = adapter_locate_front(opts["adapter"], seq.upper(), opts.get("error_rate", 0.12), opts.get("overlap", 3))
        if hit is not None:
            seq = seq[hit[3]:]
            qual = qual[hit[3]:] if qual is not None else None
        out.append(seq)
    elif opts.get("adapter"):
        hit = adapter_locate_back(opts["adapter"], seq.upper(), opts.get("error_rate", 0.12), opts.get("overlap", 3))
        if hit is not None:
            seq = seq[:hit[2]]
            qual = qual[:hit[2]] if qual is not None else None
        out.append(seq)
    if opts.get("trim_n"):
        a = 0
        while a < len(seq) and seq[a] in "Nn":
            a += 1
        b = len(seq)
        while b > a and seq[b - 1] in "Nn":
            b -= 1
        seq = seq[a:b]
        out.append(seq)
    for c in opts.get("cut", []):
        if c > 0:
            seq = seq[c:]
        elif c < 0:
            seq = seq[:c]
        if c != 0:
            out.append(seq)
    return out


def trimmed_counts(records, opts: dict, min_len: int = 16, per_modifier: bool = True):
    """The per-file dictionary the reference's worker builds (digest.py:320-375): at HEAD the length test and the count
    sit INSIDE the loop over the modifiers, so a read is counted once after every modifier (``per_modifier``); counted
    once, after the last one, is what the loop evidently meant.  -> dict in insertion order."""
    d = {}
    for seq, qual in records:
        stages = trim_stages(seq, qual, opts)
        for s in (stages if per_modifier else stages[-1:]):
            if len(s) >= min_len:
                d[s] = d.get(s, 0) + 1
    return d
