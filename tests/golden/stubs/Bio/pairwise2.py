"""Stand-in for ``Bio.pairwise2`` used ONLY by tests/golden/make_golden.py to run the reference's A-to-I code
(``mirge/libs/mirge2_tRF_a2i.py:246-295`` calls ``pairwise2.align.localms(target, read, 2, -1, -20, -20)``).

Biopython is not installed in this image, so this file re-implements the one call the reference makes, from
Biopython's documented behaviour:

* Smith-Waterman local alignment, match/mismatch scores as given, a gap of n characters costs
  ``open + (n - 1) * extend`` (here 20 per character, so a gap pays only between two segments of more than ten
  matches each: for reads the cascade annotated the result is the best UNGAPPED diagonal);
* every result is a tuple ``(seqA, seqB, score, begin, end)`` where seqA / seqB are the two FULL sequences
  written along the alignment and padded with '-' at both ends to one common length (what
  ``_finish_backtrace`` does); the reference reads only ``[0][0]`` and ``[0][1]``;
* results are ordered by the row-major position of their end cell (``_find_start`` scans rows = seqA, then
  columns = seqB), so ``[0]`` is the best-scoring alignment that ends first in seqA, then first in seqB.

The last rule and the traceback preference (diagonal before a gap) are written from memory of the Biopython
source and cannot be checked here: fixtures whose result depends on them (two different diagonals with the same
best score) are not generated -- make_golden.py asserts that the best score is reached on ONE diagonal.
"""
from collections import namedtuple

Alignment = namedtuple("Alignment", "seqA seqB score start end")


def _sw(a, b, match, mismatch, gap):
    n, m = len(a), len(b)
    H = [[0.0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        ai = a[i - 1]
        Hi, Hp = H[i], H[i - 1]
        for j in range(1, m + 1):
            d = Hp[j - 1] + (match if ai == b[j - 1] else mismatch)
            v = max(0.0, d, Hp[j] + gap, Hi[j - 1] + gap)
            Hi[j] = v
    return H


def _traceback(a, b, H, i, j, match, mismatch, gap):
    """-> (aligned a, aligned b, begin cell) for the local alignment that ends at cell (i, j)"""
    ra, rb = [], []
    while i > 0 and j > 0 and H[i][j] > 0:
        s = match if a[i - 1] == b[j - 1] else mismatch
        if H[i][j] == H[i - 1][j - 1] + s:
            ra.append(a[i - 1]); rb.append(b[j - 1]); i -= 1; j -= 1
        elif H[i][j] == H[i - 1][j] + gap:
            ra.append(a[i - 1]); rb.append("-"); i -= 1
        else:
            ra.append("-"); rb.append(b[j - 1]); j -= 1
    return "".join(reversed(ra)), "".join(reversed(rb)), i, j


class _Align:
    def localms(self, seqA, seqB, match, mismatch, open, extend):
        assert open == extend, "stand-in: linear gap cost only"
        H = _sw(seqA, seqB, match, mismatch, open)
        best = max(max(r) for r in H)
        out = []
        if best <= 0:
            return out
        for i in range(len(seqA) + 1):
            for j in range(len(seqB) + 1):
                if H[i][j] != best:
                    continue
                ma, mb, bi, bj = _traceback(seqA, seqB, H, i, j, match, mismatch, open)
                # the unaligned heads and tails, padded so that both strings have one length
                ha, hb = seqA[:bi], seqB[:bj]
                ta, tb = seqA[i:], seqB[j:]
                hl, tl = max(len(ha), len(hb)), max(len(ta), len(tb))
                fa = "-" * (hl - len(ha)) + ha + ma + ta + "-" * (tl - len(ta))
                fb = "-" * (hl - len(hb)) + hb + mb + tb + "-" * (tl - len(tb))
                out.append(Alignment(fa, fb, best, hl, hl + len(ma)))
        return out

    def diagonals_at_best(self, seqA, seqB, match, mismatch, open, extend):
        """test helper: the set of (i - j) over the end cells that reach the best score"""
        H = _sw(seqA, seqB, match, mismatch, open)
        best = max(max(r) for r in H)
        return {i - j for i in range(len(seqA) + 1) for j in range(len(seqB) + 1) if H[i][j] == best and best > 0}


align = _Align()
