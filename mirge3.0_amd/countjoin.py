"""Count join -- host mirror of ``mirge/libs/summary.py`` (:25-45, :677-798, :882-901, :1223-1291).

The sums over the U collapsed rows (per-class sums :686-698, exact / isomiR group-by :749-752,
'All miRNA Reads' :769-771) run on the GPU (``mirge_count_join``).  What is left is arithmetic on
tables with one row per miRNA (a few thousand): ``mirge_can``, the merged-family renaming, RPM
and writing ``miR.Counts.csv`` / ``miR.RPM.csv`` / ``annotation.report.csv``.  That tail is done
with pandas the way the reference does it so that dtypes -- and therefore the printed numbers
(``50`` vs ``50.0``) -- come out identical.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _ffi
from .cascade import EXACT_PASS, ISO_PASS, PASSES
from .seqio import FlatSeqs, Library

# report column <- pass index (summary.py:686-690, 895-901)
REPORT_CLASSES = [("Hairpin miRNAs", 1), ("mature tRNA Reads", 2), ("primary tRNA Reads", 3),
                  ("snoRNA Reads", 4), ("rRNA Reads", 5), ("ncRNA others", 6), ("mRNA Reads", 7)]


def _group_by_name(tab: np.ndarray, names: Sequence[str]):
    """Sum rows that carry the same reference name (pandas groupby on the name column)."""
    uniq: Dict[str, int] = {}
    idx = np.empty(len(names), dtype=np.int64)
    for i, nm in enumerate(names):
        idx[i] = uniq.setdefault(nm, len(uniq))
    out = np.zeros((len(uniq), tab.shape[1]), dtype=np.int64)
    np.add.at(out, idx, tab)
    return list(uniq), out


# annotation.report.html: pandas' table with the reference's inline styles put on every header and data cell
# (summary.py:1279-1286; its replacement of "<table>" never matches pandas' `<table border="1" class="dataframe">`)
_HTML_TH = '<th style ="background-color: #3f51b5; color:#ffffff; text-align:center">'
_HTML_TD = ('<td style="vertical-align: middle;background-color: #edf6ff;font-size: 14px;font-family: Arial;font-weight: normal;'
            'color: #000000;text-align:center; height="250";border:0; ">')


def finish_tables(class_sums: np.ndarray, exact: np.ndarray, iso: np.ndarray, mirna: Library,
                  merges: List[List[str]], base_names: List[str], sampleReadCounts, trimmedReadCounts,
                  trimmedReadCountsUnique, cr_threshold: float, spike_in: bool, workDir=None):
    """summary.py:700-798 and :882-901,1223-1291 on the per-miRNA tables.  The pandas operations follow the reference's
    one by one (same joins, fills and casts), because the printed numbers -- ``50`` or ``50.0`` -- are their dtypes."""
    import pandas as pd
    family_of: Dict[str, str] = {}    # member name -> merged (family) name, <org>_merges_<db>.csv
    universe: Dict[str, str] = {}     # every row name of miR.Counts.csv, in the reference's order: families first
    for row in merges:  # summary.py:707-712
        for member in row[1:]:
            family_of[member] = row[0]
            universe[row[0]] = "1"
    names, ex = _group_by_name(exact, mirna.names)
    _, iso_g = _group_by_name(iso, mirna.names)
    present = ex.sum(axis=1) > 0  # a name has an 'exact miRNA' row iff some read hit it in pass 0
    order = sorted(i for i in range(len(names)) if present[i])  # groupby sorts by name
    order.sort(key=lambda i: names[i])
    df = pd.DataFrame([names[i] for i in order], columns=['exact miRNA'])
    for s, file_name in enumerate(base_names):  # mirge_can, summary.py:25-45
        x = ex[order, s].astype(np.int64).copy()
        y = iso_g[order, s].astype(np.int64).copy()
        low = x < 2
        x[low] = 0
        y[low] = 0
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.where(y > 0, x / np.where(y > 0, y, 1), x.astype(np.float64))
        ok = ratio > cr_threshold
        if ok.all():
            df[file_name] = (x + y).astype(np.int64)
        else:
            df[file_name] = np.where(ok, (x + y).astype(np.float64), np.nan)
    df['miRNA'] = df['exact miRNA'].map(family_of)
    df = df.fillna(0)
    df.loc[df.miRNA == 0, 'miRNA'] = df['exact miRNA']
    df.set_index('miRNA', inplace=True)
    df = df.groupby(['miRNA']).sum()[base_names]
    filtered_reads = df.sum(axis=0, skipna=True)[base_names].to_dict()
    rpm = (df.div(df.sum(axis=0)) * 1000000).round(4)
    for h in mirna.headers:  # `bowtie-inspect -n`, summary.py:783-788
        first_word = h.split(" ")[0] if "segs:" in h else h
        if first_word not in family_of:
            universe[first_word] = "1"
    all_rows = pd.DataFrame(list(universe.keys()), columns=['miRNA'])
    all_rows.set_index('miRNA', inplace=True)
    counts_table = all_rows.join(df, how='outer').fillna(0)
    rpm_table = all_rows.join(rpm, how='outer').fillna(0)
    n_expressed = {fn: int(df.index[df[fn] > 0].shape[0]) for fn in base_names}
    classes = list(REPORT_CLASSES) + ([("Spike-in", 9)] if spike_in else [])
    all_mirna = class_sums[EXACT_PASS] + class_sums[ISO_PASS]
    report_cols = {
        'Total Input Reads': sampleReadCounts, 'Trimmed Reads (all)': trimmedReadCounts,
        'Trimmed Reads (unique)': trimmedReadCountsUnique,
        'All miRNA Reads': {fn: int(all_mirna[s]) for s, fn in enumerate(base_names)},
        'Filtered miRNA Reads': filtered_reads, 'Unique miRNAs': n_expressed,
    }
    for col, p in classes:
        report_cols[col] = {fn: int(class_sums[p][s]) for s, fn in enumerate(base_names)}
    annotated = ['All miRNA Reads'] + [c for c, _ in classes]
    column_order = ['Total Input Reads', 'Trimmed Reads (all)', 'Trimmed Reads (unique)', 'All miRNA Reads',
                    'Filtered miRNA Reads', 'Unique miRNAs'] + [c for c, _ in classes] + ['Remaining Reads']
    summary = pd.DataFrame.from_dict(report_cols).fillna(0).astype(int)
    summary['Remaining Reads'] = summary['Trimmed Reads (all)'] - (summary[annotated].sum(axis=1))
    summary = summary.reindex(columns=column_order)
    summary.index.name = "Sample name(s)"
    out = dict(counts=counts_table, rpm=rpm_table, summary=summary,
               class_sums={c: class_sums[p] for c, p in classes}, raw=(class_sums, exact, iso),
               filtered=filtered_reads)
    if workDir is not None:
        counts_table.to_csv(Path(workDir) / "miR.Counts.csv")
        rpm_table.to_csv(Path(workDir) / "miR.RPM.csv")
        summary.to_csv(Path(workDir) / "annotation.report.csv")
        html = summary.reset_index(level=['Sample name(s)'])
        html.index += 1
        with open(Path(workDir) / "annotation.report.html", 'w') as f:
            f.write(html.to_html(index=False).replace("<th>", _HTML_TH).replace("<td>", _HTML_TD))
    return out


def _calc_entropy(values) -> float:
    """``calcEntropy`` (summary.py:906-913): Shannon entropy in bits over the entries greater than ONE."""
    import math
    total = sum(values)
    e = 0
    for v in values:
        if v > 1:
            f = float(v) / total
            e = e + -1 * f * math.log(f, 2)
    return e


def isomir_entropy_tables(pdMapped, base_names, filtered_totals, workDir):
    """``-ie`` (what the reference's ``create_ie`` writes, summary.py:915-1021; outside SURVEY.md 8's rows, kept because the
    golden cases hold its two files): ``isomirs.csv`` -- one line per isomiR sequence with its RPM per sample and its
    across-sample entropy, tab separated under a comma separated header, as the reference writes it -- and
    ``isomirs.samples.csv`` -- per miRNA and sample: entropy of isomiRs + canonical, canonical share, canonical RPM, top
    isomiR RPM.  Input: the mapped rows with their 'exact miRNA' / 'isomiR miRNA' names and sample counts;
    ``filtered_totals[sample]`` = 'Filtered miRNA Reads'.  The float expressions keep the reference's operand order: the
    files are compared byte for byte."""
    import math
    samples = list(base_names)
    S = len(samples)
    table = pdMapped[samples].to_numpy(dtype=np.int64)
    sequences = [str(x) for x in pdMapped.index]
    # (a Python zero gives 0, a numpy zero goes through the division and gives inf -- as the reference's expression does)
    per_million = [0 if (type(filtered_totals[b]) is int and filtered_totals[b] == 0) else 1000000 / filtered_totals[b] for b in samples]
    h_max = math.log(S, 2)

    def family(name):  # a SNP variant counts towards its miRNA
        return name.split('.')[0] if ".SNP" in name else name

    # rows of every family: canonical rows summed per sample, isomiR rows kept one by one (in table order)
    canonical: Dict[str, np.ndarray] = {}
    for i, nm in enumerate(pdMapped['exact miRNA'].tolist()):
        if nm:
            key = family(nm)
            canonical[key] = canonical.get(key, 0) + table[i]
    variants: Dict[str, List[int]] = {}
    per_sequence = ['miRNA,sequence' + ''.join(',' + b for b in samples) + ',Entropy\n']
    for i, nm in enumerate(pdMapped['isomiR miRNA'].tolist()):
        if not nm:
            continue
        key = family(nm)
        variants.setdefault(key, []).append(i)
        row = table[i].tolist()
        spread = "NA" if h_max == 0 else str(_calc_entropy(row) / h_max)
        per_sequence.append("\t".join([key, sequences[i]] + [str(v * per_million[k]) for k, v in enumerate(row)] + [spread]) + "\n")
    per_mirna = ['miRNA' + ''.join(f',{b} isomir+miRNA Entropy,{b} Canonical Sequence,{b} Canonical RPM,{b} Top Isomir RPM' for b in samples) + '\n']
    for key, members in variants.items():
        if key not in canonical:  # isomiRs of a miRNA without canonical reads: no line (the reference's left join)
            continue
        can = canonical[key].tolist()
        cells = [key]
        for k in range(S):
            col = table[members, k].tolist()
            can_rpm = can[k] * per_million[k]
            both = can_rpm + sum(col) * per_million[k]
            cells += [str(_calc_entropy(col + [can[k]]) / math.log(len(col), 2)) if len(col) > 1 else 'NA',
                      str(100.0 * can_rpm / both) if both > 0 else 'NA', str(can_rpm), str(max(col) * per_million[k])]
        per_mirna.append(','.join(cells) + '\n')
    (Path(workDir) / "isomirs.csv").write_text("".join(per_sequence))
    (Path(workDir) / "isomirs.samples.csv").write_text("".join(per_mirna))


def summarize_device(ctx: _ffi.Context, uniq: _ffi.DeviceReads, res: _ffi.CascadeResult, mirna: Library,
                     merges, base_names, sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique,
                     cr_threshold: float = 0.1, spike_in: bool = False, workDir=None):
    """Fast path: the annotation and the count matrix never leave the GPU before the join."""
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, EXACT_PASS, ISO_PASS, len(mirna))
    return finish_tables(cls, ex, iso, mirna, merges, list(base_names), sampleReadCounts, trimmedReadCounts,
                         trimmedReadCountsUnique, cr_threshold, spike_in, workDir)


def summarize(args, workDir, ref_db, base_names, pdMapped, sampleReadCounts, trimmedReadCounts,
              trimmedReadCountsUnique):
    """Drop-in for ``summarize`` (summary.py:677): takes the mapped rows of the DataFrame.

    The name columns are turned back into (pass, reference index) arrays; those and the count
    matrix go to the GPU and ``mirge_count_join_host`` does the sums.
    """
    from .cascade import get_cascade
    from .seqio import load_merges
    casc = get_cascade(args, ref_db, getattr(args, "device", 0))
    ctx = casc.ctx
    n = len(pdMapped)
    ps = np.full(n, -1, dtype=np.int8)
    ref = np.zeros(n, dtype=np.int32)
    for p in range(casc.n_pass):
        col = PASSES[p][0]
        if col not in pdMapped.columns:
            continue
        v = pdMapped[col].to_numpy(dtype=object)
        sel = np.nonzero(v.astype(bool))[0]
        if sel.size == 0:
            continue
        lut = {}
        for i, nm in enumerate(casc.lib_of_pass(p).names):
            lut.setdefault(nm, i)
        ps[sel] = p
        ref[sel] = [lut[x] for x in v[sel]]
    counts = pdMapped[list(base_names)].to_numpy(dtype=np.int64)
    mirna = casc.libs["mirna"]
    cls, ex, iso = _ffi.count_join_host(ctx, ps, ref, counts.astype(np.uint32), casc.n_pass, EXACT_PASS, ISO_PASS,
                                        len(mirna))
    out = finish_tables(cls, ex, iso, mirna, load_merges(str(args.libraries_path), args.organism_name, ref_db),
                        list(base_names), sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique,
                        float(args.crThreshold), bool(args.spikeIn), workDir)
    if getattr(args, "isoform_entropy", False) and workDir is not None:  # -ie
        isomir_entropy_tables(pdMapped, base_names, out["filtered"], workDir)
    return out
