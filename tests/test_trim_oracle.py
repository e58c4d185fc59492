"""Known-answer tests of the oracle's restatement of cutadapt's adapter search (oracle.adapter_locate_back / _front):
the cases of cutadapt's user guide for a regular 3' adapter (-a) and a regular 5' adapter (-g), written with DNA letters.
cutadapt itself is absent from the image, so these pin the restatement to the published behaviour, not to the program
(parity of row N4 is unpinned)."""
import numpy as np

import oracle

AD = "TGGAATTCTCGG"  # 12 nt: floor(12 * 0.12) = 1 error over the full length, 0 over a part of up to 8 nt
INS = "ACGTTGCATCAGGCATTACG"  # shares no 3-mer run with the adapter's ends


def back(read, **kw):
    hit = oracle.adapter_locate_back(AD, read, kw.get("rate", 0.12), kw.get("overlap", 3))
    return read if hit is None else read[:hit[2]]


def front(read, **kw):
    hit = oracle.adapter_locate_front(AD, read, kw.get("rate", 0.12), kw.get("overlap", 3))
    return read if hit is None else read[hit[3]:]


def test_regular_three_prime_adapter_cases():
    assert back(INS + AD) == INS                       # full adapter at the end
    assert back(INS + AD + "AAAC") == INS              # full adapter inside: it and what follows go
    assert back(INS + AD[:5]) == INS                   # partial adapter at the end
    assert back(INS + AD[:2]) == INS + AD[:2]          # ... shorter than the minimum overlap: kept
    assert back(AD + INS) == ""                        # adapter at the start: nothing is left
    one = AD[:6] + "A" + AD[7:]                        # one substitution in the full adapter: within 12 % of 12
    assert back(INS + one + "CC") == INS
    two = one[:2] + "T" + one[3:]
    assert back(INS + two + "CC") == INS + two + "CC"  # two errors: not an occurrence
    assert back(INS + AD[:3] + AD[4:] + "CC") == INS   # one deleted adapter base
    assert back(INS) == INS


def test_regular_five_prime_adapter_cases():
    assert front(AD + INS) == INS                      # full adapter at the start
    assert front(AD[4:] + INS) == INS                  # partial adapter at the start (its first bases are missing)
    assert front("CCA" + AD + INS) == INS              # full adapter inside: it and what precedes it go
    assert front(AD[-2:] + INS) == AD[-2:] + INS       # two bases of overlap: below the minimum, kept
    assert front(INS + AD) == ""                       # adapter at the end: everything up to its end goes
    assert front(INS + AD[:6]) == INS + AD[:6]         # a 5' adapter must reach its last base
    one = AD[:6] + "A" + AD[7:]
    assert front(one + INS) == INS                     # one substitution over the full length
    assert front(one[5:] + INS) == one[5:] + INS       # ... but none allowed over 7 aligned bases (7 * 0.12 < 1)
    assert front("CCA" + AD[:3] + AD[4:] + INS) == INS  # one deleted adapter base
    assert front(INS) == INS


def test_first_exact_occurrence_wins_for_both_kinds():
    rng = np.random.default_rng(3)
    ad = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"
    for _ in range(200):
        a = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(5, 40))))
        b = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(5, 40))))
        read = a + ad + b + ad + a
        hit = oracle.adapter_locate_back(ad, read)
        assert hit is not None and hit[2] == len(a) and hit[4] == len(ad) and hit[5] == 0
        hit = oracle.adapter_locate_front(ad, read)
        assert hit is not None and hit[3] == len(a) + len(ad) and hit[4] == len(ad) and hit[5] == 0


def test_trim_stages_with_a_five_prime_adapter():
    read = "CC" + AD + INS + "NN"
    stages = oracle.trim_stages(read, "I" * len(read), dict(q_back=10, adapter=AD, front=True, trim_n=True, cut=[-2]))
    assert stages == [read, INS + "NN", INS, INS[:-2]]


def test_partial_adapter_ties_keep_the_longer_prefix():
    """Last column of the alignment matrix (the adapter runs off the read's end): cutadapt walks it from the longest
    adapter prefix down (`for i in reversed(range(first_i, m + 1))`) with a strict (matches, cost) improvement rule, so of
    two prefixes with equal matches and cost the LONGER one is reported.  Reads where a 16- and a 17-base prefix tie
    (one error each): the cut point is the same here, the reported adapter stretch is the longer one."""
    for read, astop, rstart in (("TCCGGCGGTATCAGACATTGGAATTCTCGGGTGGC", 17, 18), ("CTGGCGACTTCGTCGCTGGAATTCTCGGGTGCCC", 18, 16),
                                ("ACGTTACCCGGGGCTCTGGAATTCTCAG", 12, 16)):
        hit = oracle.adapter_locate_back("TGGAATTCTCGGGTGCCAAGGAACTCCAG", read)
        assert hit[1] == astop and hit[2] == rstart and hit[5] == 1, hit


def test_anchored_adapter_cases_of_the_user_guide():
    """cutadapt's user guide, "Anchored 5' adapters" / "Anchored 3' adapters" (its tables, with ADAPTER -> AD, mysequence ->
    INS): an anchored adapter is taken whole at the read's end it is anchored to, or not at all."""
    def pre(read, rate=0.12):
        hit = oracle.adapter_locate_front(AD, read, rate, 3, anchored=True)
        return read if hit is None else read[hit[3]:]

    def suf(read, rate=0.12):
        hit = oracle.adapter_locate_back(AD, read, rate, 3, anchored=True)
        return read if hit is None else read[:hit[2]]
    assert pre(AD + INS) == INS                          # ADAPTERmysequence -> mysequence
    assert pre(AD[1:] + INS, rate=0.0) == AD[1:] + INS   # DAPTERmysequence: no partial occurrence (it would be one deletion)
    assert pre("C" + AD + INS, rate=0.0) == "C" + AD + INS  # something in front of it: not at the first base
    assert pre(INS + AD) == INS + AD                     # mysequenceADAPTER: not a 5' occurrence at all
    one = AD[:6] + "A" + AD[7:]
    assert pre(one + INS) == INS                         # one substitution within 12 % of 12
    assert pre(AD[:3] + AD[4:] + INS) == INS             # one deleted adapter base: an error like any other
    assert pre("G" + AD + INS) == INS                    # one inserted read base in front: one error, the prefix goes with it
    assert suf(INS + AD) == INS                          # mysequenceADAPTER -> mysequence
    assert suf(INS + AD[:8]) == INS + AD[:8]             # mysequenceADAP: a partial adapter is no anchored occurrence
    assert suf(INS + AD + "ACGT") == INS + AD + "ACGT"   # mysequenceADAPTERsomethingelse: not at the read's end
    assert suf(INS + one) == INS
    assert suf(AD) == ""


def test_linked_adapter_cases_of_the_user_guide():
    """"Linked adapters (combined 5' and 3' adapter)": `-a ADAPTER1...ADAPTER2` -- ADAPTER1 anchored and required, ADAPTER2
    optional; `-g ADAPTER1...ADAPTER2` -- ADAPTER1 a regular 5' adapter, both required (a read without either stays)."""
    A1, A2 = "TTAGGCAC", AD

    def linked(read, kind, **kw):
        lk = dict(front=A1, back=A2, front_anchored=kind == "back", back_anchored=False, front_required=True, back_required=kind == "front")
        lk.update(kw)
        return oracle.trim_stages(read, None, dict(linked=lk))[-1]
    # -a: both found; only the 5' part (3' optional: still trimmed); 5' part absent (required): untouched, even with the 3' part there
    assert linked(A1 + INS + A2, "back") == INS
    assert linked(A1 + INS + A2 + "ACG", "back") == INS
    assert linked(A1 + INS, "back") == INS
    assert linked(INS + A2, "back") == INS + A2
    assert linked("C" + A1 + INS + A2, "back", ) == "C" + A1 + INS + A2   # (0 errors allowed over 8 bases: not anchored at base 1)
    # -g: regular 5' part anywhere in front, both required
    assert linked(A1 + INS + A2, "front") == INS
    assert linked("CCA" + A1 + INS + A2, "front") == INS
    assert linked(A1[3:] + INS + A2, "front") == INS                      # a partial 5' part at the read's start
    assert linked(A1 + INS, "front") == A1 + INS                         # the 3' part is required: nothing is removed
    assert linked(INS + A2, "front") == INS + A2                         # the 5' part is required
    # the 3' part is searched BEHIND the 5' match only: an A2 inside what the 5' part removes does not count
    assert linked(A2[:6] + A1 + INS, "back", front_anchored=False) == INS
    # explicit anchors of a -g linked adapter
    assert linked(A1 + INS + A2, "front", front_anchored=True, back_anchored=True) == INS
    assert linked(A1 + INS + A2 + "ACA", "front", front_anchored=True, back_anchored=True) == A1 + INS + A2 + "ACA"  # (one extra base would be one error: allowed)
    # the documented command line's adapter (docs/source/quick_start.md:213-220)
    lk = dict(front="TTAGGC", back="TGGAATTCTCGGGTGCCAAGGAACTCCAGT", front_anchored=False, back_anchored=False, front_required=True, back_required=True)
    read = "TTAGGC" + "TGAGGTAGTAGGTTGTATAGTT" + "TGGAATTCTCGGGTGCCAAGG"
    assert oracle.trim_stages(read, "I" * len(read), dict(q_back=10, linked=lk)) == [read, "TGAGGTAGTAGGTTGTATAGTT"]
