"""Per-position variant tally and the A-to-I table (BASELINE config 5; SURVEY.md 8 row a16, 'next' N1).

Scope of this module: the counting core of ``A2IEditing`` / ``judgeAllign``
(``mirge/libs/mirge2_tRF_a2i.py:298-419``) on the GPU (``mirge_variant_tally``) and the statistics the
reference derives from those counts (ratio :405-407, binomial p-value :408-411, Benjamini-Hochberg
:1175-1183).  NOT built: the report plumbing around it (RPM gates :1115-1128, the genome-uniqueness
filter that bowtie-aligns the reads against the whole genome :1056-1096, SNP-pseudo FASTA, CSV layout).

Parity status: **unpinned**.  The reference aligns every read to the canonical sequence with
``Bio.pairwise2.align.localms(target, read, 2, -1, -20, -20)`` (an ungapped best local alignment;
Biopython is absent here, so no golden vector can be produced); this build uses the cascade's own
ungapped alignment of the read to its miRNA (same diagonal whenever the cascade's hit is the best
local alignment, which it is for <= 2 mismatches in >= 13 aligned bases).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from . import _ffi
from .cascade import Cascade, EXACT_PASS, ISO_PASS, PASSES

P_MISMATCH = 0.001  # mirge2_tRF_a2i.py:336
TAIL_SHIFT = 5      # :337
A, G = 0, 2


def tally(casc: Cascade, uniq: _ffi.DeviceReads, res: _ffi.CascadeResult):
    """-> (accepted [R,S], canonical [R,S], census [R,32,4,4,S]) for the cascade's miRNA library."""
    trim5 = PASSES[ISO_PASS][3].get("trim5", 0)
    return _ffi.variant_tally(casc.ctx, uniq, res, casc._dev["mirna"], EXACT_PASS, ISO_PASS, trim5)


def a_to_i_table(accepted: np.ndarray, census: np.ndarray, mirna_names: List[str], mirna_lens: np.ndarray,
                 samples: List[str]) -> List[Dict]:
    """Rows (miRNA, 1-based position, per-sample count / total / ratio / p / BH-adjusted p) for every
    canonical 'A' position q < len-5 that some accepted read shows as 'G' (:358-366, :405-413, :1175-1183)."""
    from scipy import stats
    R, S = accepted.shape
    rows = []
    for r in range(R):
        Lc = int(mirna_lens[r])
        for q in range(0, max(Lc - TAIL_SHIFT, 0)):
            ag = census[r, q, A, G, :]
            if not ag.any():
                continue
            row = dict(miRNA=mirna_names[r], position=q + 1, count={}, total={}, ratio={}, p_value={})
            for s, nm in enumerate(samples):
                tot, c = int(accepted[r, s]), int(ag[s])
                if c == 0:
                    continue
                row["count"][nm], row["total"][nm] = c, tot
                row["ratio"][nm] = c / tot if tot else 0.0
                row["p_value"][nm] = float(stats.binom.cdf(tot - c, tot, 1 - P_MISMATCH)) if tot - c >= 0 else 1.0
            rows.append(row)
    for nm in samples:  # Benjamini-Hochberg per sample, as the reference writes it (:1175-1183)
        plist = sorted((row["p_value"][nm], i) for i, row in enumerate(rows) if nm in row["p_value"])
        for rank, (p, i) in enumerate(plist):
            rows[i].setdefault("adjusted_p", {})[nm] = p * len(plist) / (rank + 1)
    return rows


def mismatch_census(census: np.ndarray) -> np.ndarray:
    """[12, S]: count-weighted totals of the 12 base changes (canonical -> read), A>C A>G A>T C>A ..."""
    out = []
    for cb in range(4):
        for rb in range(4):
            if cb != rb:
                out.append(census[:, :, cb, rb, :].sum(axis=(0, 1)))
    return np.stack(out)
