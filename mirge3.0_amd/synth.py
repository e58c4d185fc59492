"""Seeded synthetic small-RNA libraries and reads.

No real library ships with the reference (they are downloaded bowtie indexes), so the
tests and the bench use libraries modelled on the human set: sizes, length ranges and the
class mix of the reads follow SURVEY.md 8(d).  Everything is vectorised numpy so that a
10 M-read sample is generated in well under a minute.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from .seqio import FlatSeqs, Library

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

# name -> (n_refs, min_len, max_len) per scale
SCALES: Dict[str, Dict[str, Tuple[int, int, int]]] = {
    # few kB: golden fixtures driven through the reference harness
    "tiny": dict(mirna=(60, 18, 25), hairpin=(45, 60, 110), mature_trna=(12, 70, 90),
                 pre_trna=(12, 90, 120), snorna=(15, 60, 200), rrna=(3, 120, 900),
                 ncrna_others=(20, 100, 400), mrna=(20, 300, 1200)),
    # CI: small enough for the brute-force oracle
    "ci": dict(mirna=(600, 18, 25), hairpin=(420, 60, 120), mature_trna=(100, 70, 90),
               pre_trna=(100, 90, 120), snorna=(200, 60, 250), rrna=(6, 120, 5000),
               ncrna_others=(1500, 100, 1000), mrna=(600, 500, 6000)),
    # SURVEY 8(d) 'small-mrna' variant: human-sized small libraries, 2 Mb mRNA
    "small": dict(mirna=(2656, 18, 25), hairpin=(1917, 60, 120), mature_trna=(430, 70, 90),
                  pre_trna=(430, 90, 120), snorna=(950, 60, 250), rrna=(8, 120, 13000),
                  ncrna_others=(4000, 100, 1000), mrna=(620, 500, 6000)),
    # human-sized: ~8 Mb ncRNA, ~130 Mb mRNA
    "full": dict(mirna=(2656, 18, 25), hairpin=(1917, 60, 120), mature_trna=(430, 70, 90),
                 pre_trna=(430, 90, 120), snorna=(950, 60, 250), rrna=(8, 120, 13000),
                 ncrna_others=(20000, 100, 1000), mrna=(40000, 500, 6000)),
}


def _rand_flat(rng: np.random.Generator, lens: np.ndarray) -> FlatSeqs:
    offsets = np.zeros(lens.shape[0] + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    data = ACGT[rng.integers(0, 4, size=int(offsets[-1]), dtype=np.uint8)]
    return FlatSeqs(data, offsets)


@dataclass
class SynthLibs:
    libs: Dict[str, Library]
    merges: List[List[str]]
    # where each mature miRNA sits in its hairpin (for templated isomiRs)
    mir_hairpin: np.ndarray
    mir_hairpin_off: np.ndarray
    # repeats=True: the consensus sequences of the Alu-like families and the simple-repeat units written into the libraries
    repeat_families: List[str] = None
    repeat_units: List[str] = None


REPEAT_UNITS = ["CA", "GT", "AT", "AAT", "CAG", "GGAA", "TTTA", "A", "T"]


def _mutated_copy(rng, cons: np.ndarray, div: float) -> np.ndarray:
    """a copy of ``cons`` with a share ``div`` of its bases substituted"""
    out = cons.copy()
    hit = np.nonzero(rng.random(out.shape[0]) < div)[0]
    code = np.searchsorted(ACGT, out[hit])
    out[hit] = ACGT[(code + rng.integers(1, 4, size=hit.shape[0])) % 4]
    return out


def _add_repeat_structure(rng, libs: Dict[str, Library]):
    """The repeat structure real libraries have and uniform draws lack (round 5's review, 'missing 4'; the reference's libraries
    are Ensembl transcripts and ncRNA, mirge/libs/bamFmt.py:13,34): poly-A tails on most mRNAs, a few 300-nt Alu-like families at
    5-15 % divergence in ~10 % of the transcripts, simple repeats ((CA)n, (GT)n, (AAT)n ...) in ncRNA and mRNA.  In place, lengths
    unchanged (stretches are overwritten); -> (family consensus sequences, repeat units)."""
    fams = [ACGT[rng.integers(0, 4, size=300, dtype=np.uint8)] for _ in range(5)]
    # Alu-likes share a poly-A linker and an A-rich tail, as the real ones do
    for f in fams:
        f[120:135] = ord("A")
        f[280:300] = ord("A")
    for key in ("mrna", "ncrna_others"):
        fs = libs[key].seqs
        lens = fs.lengths
        n = len(fs)
        if key == "mrna":  # poly-A tails: 60 % of the transcripts, 20-150 nt
            for i in np.nonzero(rng.random(n) < 0.6)[0]:
                t = int(min(rng.integers(20, 151), lens[i] // 3))
                fs.data[fs.offsets[i + 1] - t:fs.offsets[i + 1]] = ord("A")
        for i in np.nonzero((rng.random(n) < 0.10) & (lens >= 360))[0]:  # one Alu-like element, 5-15 % diverged
            cp = _mutated_copy(rng, fams[int(rng.integers(0, len(fams)))], float(rng.uniform(0.05, 0.15)))
            at = int(fs.offsets[i] + rng.integers(10, lens[i] - 320))
            fs.data[at:at + 300] = cp
        share = 0.05 if key == "ncrna_others" else 0.03
        for i in np.nonzero((rng.random(n) < share) & (lens >= 300))[0]:  # a simple repeat of 30-200 nt
            unit = np.frombuffer(REPEAT_UNITS[int(rng.integers(0, 7))].encode(), dtype=np.uint8)
            t = int(min(rng.integers(30, 201), lens[i] // 2))
            at = int(fs.offsets[i] + rng.integers(5, lens[i] - t - 5))
            fs.data[at:at + t] = np.resize(unit, t)
    return ["".join(chr(c) for c in f) for f in fams], list(REPEAT_UNITS)


def make_libraries(seed: int = 20260101, scale: str = "small", repeats: bool = False) -> SynthLibs:
    """``repeats=True``: the same libraries with the repeat structure of real ones written in (``_add_repeat_structure``; tRNA
    isodecoder families: identical bodies, 1-3 differing bases).  Drawn from a generator of its own: ``repeats=False`` is byte for
    byte what it always was."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rng_r = np.random.Generator(np.random.PCG64([seed, 0x7265]))
    spec = SCALES[scale]
    libs: Dict[str, Library] = {}

    # ---- mature miRNA: base set + ~10 % one-base 'SNP' variants + a few 3'-variant families
    n_mir, lo, hi = spec["mirna"]
    n_base = max(4, int(round(n_mir / 1.15)))
    base_lens = rng.integers(lo, hi + 1, size=n_base)
    base = _rand_flat(rng, base_lens)
    seqs = base.to_list()
    names = [f"hsa-miR-{i + 1}-{'5p' if i % 2 == 0 else '3p'}" for i in range(n_base)]
    n_snp = max(1, int(round(0.10 * n_base)))
    for j in range(n_snp):
        src = int(rng.integers(0, n_base))
        s = bytearray(seqs[src].encode())
        p = int(rng.integers(2, len(s) - 2))
        s[p] = ACGT[(int(np.searchsorted(ACGT, s[p])) + int(rng.integers(1, 4))) % 4]
        seqs.append(s.decode())
        names.append(f"{names[src]}-SNP{j + 1}")
    while len(seqs) < n_mir:  # family members: same 5' body, different last two bases
        src = int(rng.integers(0, n_base))
        s = seqs[src][:-2] + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=2))
        seqs.append(s)
        names.append(f"hsa-let-{len(seqs)}")
    libs["mirna"] = Library(names, FlatSeqs.from_list(seqs))

    # ---- hairpins: random precursor with its mature(s) written in
    n_hp, lo, hi = spec["hairpin"]
    hp_lens = rng.integers(lo, hi + 1, size=n_hp)
    hp = _rand_flat(rng, hp_lens)
    mir_hp = np.full(len(seqs), -1, dtype=np.int64)
    mir_off = np.zeros(len(seqs), dtype=np.int64)
    for i in range(len(seqs)):
        h = i % n_hp
        arm5 = (i // n_hp) % 2 == 0
        L = len(seqs[i])
        off = 8 + int(rng.integers(0, 4)) if arm5 else int(hp_lens[h]) - L - 8 - int(rng.integers(0, 4))
        s, e = hp.offsets[h] + off, hp.offsets[h] + off + L
        hp.data[s:e] = np.frombuffer(seqs[i].encode(), dtype=np.uint8)
        mir_hp[i], mir_off[i] = h, off
    # later writes may overwrite earlier matures in the same hairpin; that only turns some
    # 'exact' reads into hairpin-less ones, which the oracle classifies all the same
    libs["hairpin"] = Library([f"hsa-mir-{i + 1}" for i in range(n_hp)], hp)

    # ---- tRNA
    n_t, lo, hi = spec["mature_trna"]
    t_lens = rng.integers(lo, hi + 1, size=n_t)
    mt = _rand_flat(rng, t_lens)
    for i in range(n_t):
        mt.data[mt.offsets[i + 1] - 3:mt.offsets[i + 1]] = np.frombuffer(b"CCA", dtype=np.uint8)
    if repeats:  # isodecoder families of ~6: the body of the family's first member, 1-3 bases changed, 'CCA' kept
        for i in range(n_t):
            head = i - i % 6
            if i == head:
                continue
            L = int(min(t_lens[i], t_lens[head]))
            mt.data[mt.offsets[i]:mt.offsets[i] + L - 3] = mt.data[mt.offsets[head]:mt.offsets[head] + L - 3]
            for p in rng_r.integers(0, L - 3, size=int(rng_r.integers(1, 4))):
                old = mt.data[mt.offsets[i] + p]
                mt.data[mt.offsets[i] + p] = ACGT[(int(np.searchsorted(ACGT, old)) + int(rng_r.integers(1, 4))) % 4]
    libs["mature_trna"] = Library([f"tRNA-{i + 1}-mature" for i in range(n_t)], mt)
    pre_seqs = []
    mts = mt.to_list()
    for i in range(spec["pre_trna"][0]):
        body = mts[i % n_t][:-3]
        lead = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(5, 16))))
        trail = "".join("ACG"[int(x)] for x in rng.integers(0, 3, size=int(rng.integers(8, 20))))
        pre_seqs.append(lead + body + trail)
    libs["pre_trna"] = Library([f"tRNA-{i + 1}-pre" for i in range(len(pre_seqs))],
                               FlatSeqs.from_list(pre_seqs))

    # ---- snoRNA, rRNA, other ncRNA, mRNA
    for key, prefix in (("snorna", "SNORD"), ("rrna", "RNA45S"), ("ncrna_others", "ENST0000N"),
                        ("mrna", "ENST0000")):
        n, lo, hi = spec[key]
        if key == "rrna":
            lens = np.linspace(lo, hi, n).astype(np.int64)
        else:
            lens = rng.integers(lo, hi + 1, size=n)
        fs = _rand_flat(rng, lens)
        names_k = [f"{prefix}{i + 1:06d}" for i in range(n)]
        headers = list(names_k)
        if key in ("mrna", "ncrna_others"):
            # header shape of the Ensembl-derived libraries (mirge/libs/bamFmt.py:13,34):
            # bowtie reports only the first token as RNAME
            headers = [f"{nm} chr{1 + i % 22} segs:{100 + i}-{100 + i + int(lens[i])} cds:+:1"
                       for i, nm in enumerate(names_k)]
        if key == "ncrna_others":
            # a few ambiguous reference bases: windows over them are never valid hits
            for i in range(0, n, max(1, n // 7)):
                p = fs.offsets[i] + int(lens[i]) // 2
                fs.data[p:p + 3] = ord("N")
        # shared 40-mers across references of the same class create multi-reference ties
        for _ in range(max(1, n // 50)):
            a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
            if a == b or lens[a] < 60 or lens[b] < 60:
                continue
            pa = fs.offsets[a] + int(rng.integers(0, lens[a] - 40))
            pb = fs.offsets[b] + int(rng.integers(0, lens[b] - 40))
            fs.data[pb:pb + 40] = fs.data[pa:pa + 40]
        libs[key] = Library(names_k, fs, headers)

    # ---- merged families (mirge/libs/summary.py:707-712)
    merges: List[List[str]] = []
    n_fam = min(50, n_base // 4)
    for f in range(n_fam):
        a, b = names[2 * f], names[2 * f + 1]
        merges.append([f"{a}/{b.split('-', 2)[-1]}", a, b])
    fams = units = None
    if repeats:
        fams, units = _add_repeat_structure(rng_r, libs)
    return SynthLibs(libs, merges, mir_hp, mir_off, fams, units)


# class mix of the raw reads
DEFAULT_MIX = dict(exact=0.45, isomir=0.15, hairpin=0.03, mature_trna=0.08, pre_trna=0.01,
                   snorna=0.04, rrna=0.08, ncrna_others=0.03, mrna=0.05, random=0.08)
# the same sample from repeat-rich libraries (make_libraries(repeats=True)): 15 % of the reads are what such libraries attract --
# poly-A / poly-T, simple-repeat and Alu-derived reads, each with 0-2 errors
REPEAT_MIX = dict(exact=0.40, isomir=0.13, hairpin=0.02, mature_trna=0.08, pre_trna=0.01, snorna=0.03, rrna=0.07, ncrna_others=0.03,
                  mrna=0.04, random=0.04, poly=0.06, simple=0.03, alu=0.06)
MAXLEN = 50


def _substrings(rng, lib: Library, m: int, lens: np.ndarray):
    """m random substrings of ``lib`` of the wanted lengths -> (matrix [m, MAXLEN], lens)."""
    reflen = lib.seqs.lengths
    r = rng.integers(0, len(lib), size=m)
    ln = np.minimum(lens, reflen[r])
    start = np.floor(rng.random(m) * (reflen[r] - ln + 1)).astype(np.int64)
    base = lib.seqs.offsets[:-1][r] + start
    return _gather(lib.seqs.data, base, ln), ln


def _gather(data: np.ndarray, base: np.ndarray, ln: np.ndarray) -> np.ndarray:
    cols = np.arange(MAXLEN, dtype=np.int64)[None, :]
    idx = np.minimum(base[:, None] + cols, data.shape[0] - 1)
    mat = data[idx]
    mat[cols >= ln[:, None]] = 0
    return mat


def _mutate(rng, mat: np.ndarray, ln: np.ndarray, n_sub: np.ndarray) -> None:
    """In place: ``n_sub[i]`` (0..2) substitutions at distinct random positions of row i."""
    m = mat.shape[0]
    rows = np.arange(m)
    p1 = np.floor(rng.random(m) * ln).astype(np.int64)
    p2 = (p1 + 1 + np.floor(rng.random(m) * np.maximum(ln - 1, 1)).astype(np.int64)) % np.maximum(ln, 1)
    for k, p in ((1, p1), (2, p2)):
        sel = n_sub >= k
        if not sel.any():
            continue
        old = mat[rows[sel], p[sel]]
        code = np.searchsorted(ACGT, old)  # A,C,G,T are sorted ASCII
        new = ACGT[(code + rng.integers(1, 4, size=code.shape[0])) % 4]
        mat[rows[sel], p[sel]] = new


def make_reads(sl: SynthLibs, n: int, seed: int = 1, mix: Dict[str, float] | None = None,
               long_frac: float = 0.05, n_frac: float = 0.001,
               pool: int | None = None, zipf_s: float = 1.1) -> FlatSeqs:
    """``n`` raw (already trimmed) reads.

    ``pool=None``: every read is drawn independently (duplicates arise naturally from the
    exact-miRNA class).  ``pool=P``: P templates are drawn, then n reads are sampled from
    them with Zipf(s) weights -- the 'realistic' U/N of SURVEY.md 8(d).
    """
    rng = np.random.Generator(np.random.PCG64([seed, 0x6d69]))
    if pool is not None:
        tmpl = make_reads(sl, pool, seed=seed + 7919, mix=mix, long_frac=long_frac, n_frac=n_frac)
        w = 1.0 / np.arange(1, pool + 1, dtype=np.float64) ** zipf_s
        pick = rng.choice(pool, size=n, p=w / w.sum())
        return tmpl.take(pick)
    mix = dict(DEFAULT_MIX if mix is None else mix)
    tot = sum(mix.values())
    keys = list(mix)
    counts = np.floor(np.array([mix[k] / tot for k in keys]) * n).astype(np.int64)
    counts[0] += n - counts.sum()
    parts: List[np.ndarray] = []
    part_lens: List[np.ndarray] = []
    L = sl.libs
    for key, m in zip(keys, counts):
        m = int(m)
        if m == 0:
            continue
        if key == "exact":
            mir = L["mirna"]
            r = rng.integers(0, len(mir), size=m)
            ln = mir.seqs.lengths[r].copy()
            start = np.zeros(m, dtype=np.int64)
            trunc = rng.random(m) < 0.10  # truncated but still exact substrings
            cut = rng.integers(0, 3, size=m)
            ln2 = np.maximum(ln - cut, 16)
            ln = np.where(trunc, np.minimum(ln, ln2), ln)
            mat = _gather(mir.seqs.data, mir.seqs.offsets[:-1][r] + start, ln)
        elif key == "isomir":
            mir, hp = L["mirna"], L["hairpin"]
            r = rng.integers(0, len(mir), size=m)
            h, off = sl.mir_hairpin[r], sl.mir_hairpin_off[r]
            d5 = rng.choice([-1, 0, 0, 0, 1], size=m)
            d3 = rng.choice([-2, -1, 0, 0, 1, 2], size=m)
            s = np.maximum(off + d5, 0)
            e = np.minimum(off + mir.seqs.lengths[r] + d3, hp.seqs.lengths[h])
            ln = np.clip(e - s, 16, MAXLEN - 2)
            mat = _gather(hp.seqs.data, hp.seqs.offsets[:-1][h] + s, ln)
            _mutate(rng, mat, ln, rng.choice([0, 1, 2], size=m, p=[0.5, 0.35, 0.15]))
            add = rng.random(m) < 0.10  # non-templated 3' addition
            rows = np.nonzero(add)[0]
            mat[rows, ln[rows]] = np.where(rng.random(rows.shape[0]) < 0.5, ord("A"), ord("T"))
            ln = ln + add
        elif key == "pre_trna":
            pre = L["pre_trna"]
            r = rng.integers(0, len(pre), size=m)
            body = rng.integers(13, 27, size=m)
            reflen = pre.seqs.lengths[r]
            body = np.minimum(body, reflen)
            mat = _gather(pre.seqs.data, pre.seqs.offsets[:-1][r] + reflen - body, body)
            tails = rng.integers(3, 7, size=m)
            cols = np.arange(MAXLEN)[None, :]
            tmask = (cols >= body[:, None]) & (cols < (body + tails)[:, None])
            mat[tmask] = ord("T")
            ln = body + tails
        elif key == "random":
            ln = rng.integers(16, 31, size=m)
            mat = ACGT[rng.integers(0, 4, size=(m, MAXLEN), dtype=np.uint8)]
            mat[np.arange(MAXLEN)[None, :] >= ln[:, None]] = 0
        elif key in ("poly", "simple"):  # homopolymer / simple-repeat reads with 0-2 errors (REPEAT_MIX)
            ln = rng.integers(16, 31, size=m)
            units = ["A", "A", "A", "T"] if key == "poly" else [u for u in (sl.repeat_units or REPEAT_UNITS) if len(u) > 1]
            which = rng.integers(0, len(units), size=m)
            phase = rng.integers(0, 4, size=m)
            mat = np.zeros((m, MAXLEN), dtype=np.uint8)
            cols = np.arange(MAXLEN, dtype=np.int64)[None, :]
            for ui, u in enumerate(units):
                rows = np.nonzero(which == ui)[0]
                ub = np.frombuffer(u.encode(), dtype=np.uint8)
                mat[rows] = ub[(cols + phase[rows][:, None]) % len(u)]
            mat[cols >= ln[:, None]] = 0
            _mutate(rng, mat, ln, rng.choice([0, 1, 2], size=m, p=[0.5, 0.3, 0.2]))
        elif key == "alu":  # reads out of the Alu-like families' consensus sequences, 0-2 errors on top of the copies' divergence
            fams = sl.repeat_families or ["".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=300))]
            cons = FlatSeqs.from_list(fams)
            want = rng.integers(16, 31, size=m)
            mat, ln = _substrings(rng, Library([f"alu{i}" for i in range(len(fams))], cons), m, want)
            _mutate(rng, mat, ln, rng.choice([0, 1, 2], size=m, p=[0.4, 0.35, 0.25]))
        else:
            lib = L[key]
            if key == "hairpin":
                want = rng.integers(26, 36, size=m)
            else:
                want = rng.integers(16, 31, size=m)
                lng = rng.random(m) < long_frac * 2.0
                want = np.where(lng, rng.integers(31, MAXLEN + 1, size=m), want)
            mat, ln = _substrings(rng, lib, m, want)
            _mutate(rng, mat, ln, rng.choice([0, 1, 2], size=m, p=[0.7, 0.2, 0.1]))
        parts.append(mat)
        part_lens.append(np.asarray(ln, dtype=np.int64))
    mat = np.concatenate(parts, axis=0)
    ln = np.concatenate(part_lens)
    # ambiguous base calls
    nn = rng.random(mat.shape[0]) < n_frac
    rows = np.nonzero(nn)[0]
    mat[rows, np.floor(rng.random(rows.shape[0]) * ln[rows]).astype(np.int64)] = ord("N")
    perm = rng.permutation(mat.shape[0])
    mat, ln = mat[perm], ln[perm]
    offsets = np.zeros(mat.shape[0] + 1, dtype=np.int64)
    np.cumsum(ln, out=offsets[1:])
    data = mat[np.arange(MAXLEN)[None, :] < ln[:, None]]
    return FlatSeqs(np.ascontiguousarray(data), offsets)


def make_reads_chunked(sl: SynthLibs, n: int, seed: int = 1, chunk: int = 2_000_000,
                       **kw) -> FlatSeqs:
    """Same distribution as ``make_reads`` with bounded peak memory for multi-million n."""
    datas, lens = [], []
    done, c = 0, 0
    while done < n:
        m = min(chunk, n - done)
        fs = make_reads(sl, m, seed=seed * 1000 + c, **kw)
        datas.append(fs.data)
        lens.append(fs.lengths)
        done += m
        c += 1
    ln = np.concatenate(lens)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(ln, out=offsets[1:])
    return FlatSeqs(np.concatenate(datas), offsets)
