"""Differential fuzzing on the GPU: random libraries (sizes chosen so that the probe length K and the
plain / recursive plans vary), random reads (substrings with 0-3 substitutions, N calls, T tails, junk,
1-255 nt, and 256-900 nt: the long class) and RANDOM cascades (any -n / -v budget 0..3, -5/-3 trims, length filters, T-tail rule) through
mirge_cascade_run, against the brute-force oracle given the same policies.  Bit-exact on every field."""
import numpy as np
import pytest

import oracle
import mirge3_amd  # noqa: F401
from mirge3_amd import _ffi
from mirge3_amd.seqio import FlatSeqs

pytestmark = pytest.mark.gpu


def _rand_seq(rng, L, pn=0.0):
    s = rng.integers(0, 4, size=L)
    out = np.frombuffer(b"ACGT", np.uint8)[s].copy()
    if pn > 0:
        out[rng.random(L) < pn] = ord("N")
    return out.tobytes().decode()


def _policy(rng, exact_bias=False):
    mode = int(rng.integers(0, 2))
    mm = int(rng.choice([0, 1, 1, 2, 2, 3]))
    if exact_bias and rng.random() < 0.65:
        # the passes one whole-read lookup answers (kernels_cascade.hpp, ExactStep): "-v 0", or "-n 0" with a length rule that
        # keeps the read inside the seed -- and their near misses: a rule one base beyond the seed, a shorter seed
        mm = 0
        pol = dict(mode=mode, mm=0, seedlen=int(rng.choice([28, 28, 28, 20])), maxtotal=(int(rng.integers(0, 3)) if mode == 0 else 0))
        if mode == 0 and rng.random() < 0.8:
            pol["len_lt"] = int(rng.integers(16, pol["seedlen"] + 3))
        elif rng.random() < 0.3:
            pol["len_lt"] = int(rng.integers(18, 40))
        if rng.random() < 0.35:
            pol["ttail"] = 1
        if rng.random() < 0.2:
            pol["trim5"] = int(rng.integers(0, 3)); pol["trim3"] = int(rng.integers(0, 3))
        return pol
    pol = dict(mode=mode, mm=mm, seedlen=int(rng.choice([28, 28, 20, 12])), maxtotal=(int(rng.integers(mm, 4)) if mode == 0 else mm))
    if rng.random() < 0.3:
        pol["trim5"] = int(rng.integers(0, 3)); pol["trim3"] = int(rng.integers(0, 3))
    if rng.random() < 0.15:
        pol["ttail"] = 1
    if rng.random() < 0.15:
        pol["len_lt"] = int(rng.integers(18, 40))
    elif rng.random() < 0.15:
        pol["len_gt"] = int(rng.integers(10, 30))
    return pol


# MIRGE_FUZZ_SEEDS=N widens the sweep for a one-off soak (default 16 cascades)
import os  # noqa: E402
N_FUZZ = int(os.environ.get("MIRGE_FUZZ_SEEDS", "16"))


@pytest.mark.parametrize("seed", range(N_FUZZ))
def test_random_cascade_matches_bruteforce(seed):
    _random_cascade(seed, False)


@pytest.mark.parametrize("seed", range(N_FUZZ))
def test_random_cascade_with_exact_passes_matches_bruteforce(seed):
    """the same with most policies drawn from the exact family: the passes k_cascade_bulk answers with one whole-read lookup
    inside a neighbouring pass's walk (pre / post steps), in every order and mixture with alignment passes"""
    _random_cascade(seed, True)


def _random_cascade(seed, exact_bias):
    rng = np.random.default_rng((5000 if exact_bias else 1000) + seed)
    ctx = _ffi.Context(0)
    n_pass = int(rng.integers(2, 9))
    libs, dev, pols = [], [], []
    same_policy_run = rng.random() < 0.5  # consecutive identical policies -> merged launch
    for p in range(n_pass):
        nref = int(rng.choice([3, 20, 150, 1200]))
        lo, hi = [(18, 25), (40, 120), (100, 600), (250, 900)][int(rng.integers(0, 4))]
        seqs = [_rand_seq(rng, int(rng.integers(lo, hi + 1)), pn=0.003 if rng.random() < 0.3 else 0.0) for _ in range(nref)]
        if rng.random() < 0.3:  # repeats: long buckets
            seqs += [("A" * int(rng.integers(20, 200))) + _rand_seq(rng, 30), "CA" * int(rng.integers(15, 80))]
        fs = FlatSeqs.from_list(seqs)
        libs.append(fs)
        dev.append(_ffi.DeviceLibrary(ctx, fs))
        if p > 0 and same_policy_run and rng.random() < 0.6:
            pols.append(dict(pols[-1]))
        else:
            pols.append(_policy(rng, exact_bias))
    reads = []
    for _ in range(3000):
        lib = libs[int(rng.integers(0, n_pass))]
        s = lib.get(int(rng.integers(0, len(lib))))
        kind = rng.random()
        u = rng.random()
        L = int(rng.integers(256, 900)) if u < 0.03 else (int(rng.integers(1, 256)) if u < 0.12 else int(rng.integers(12, 45)))
        if kind < 0.75 and len(s) > 2:
            L = min(L, len(s))
            a = int(rng.integers(0, len(s) - L + 1))
            x = list(s[a:a + L])
            for _ in range(int(rng.choice([0, 0, 0, 0, 1, 2] if exact_bias else [0, 0, 1, 1, 2, 3]))):
                q = int(rng.integers(0, L)); x[q] = "ACGT"[int(rng.integers(0, 4))]
            if rng.random() < 0.05:
                x[int(rng.integers(0, L))] = "N"
            r = "".join(x)
            if rng.random() < 0.1:
                r = (r + "T" * int(rng.integers(3, 8)))[:1000]
            if rng.random() < 0.1:
                r = ("ACGT"[int(rng.integers(0, 4))] + r + _rand_seq(rng, 2))[:1000]
        else:
            r = _rand_seq(rng, L, pn=0.02)
        reads.append(r)
    fs = FlatSeqs.from_list(reads)
    policies = []
    for pol in pols:
        p = _ffi.MirgePolicy()
        for k, v in pol.items():
            setattr(p, k, v)
        policies.append(p)
    dr = _ffi.DeviceReads.pack(ctx, fs)
    res = _ffi.cascade_run(ctx, dr, dev, policies)
    g = res.fetch()
    # oracle: passes >= 1 see only unannotated rows; pass 0 sees everything (nothing is annotated yet)
    opol = [dict(pol, need_unannotated=1) for pol in pols]
    o = oracle.cascade(fs.data, fs.offsets, [(l.data, l.offsets) for l in libs], n_pass=n_pass, indexed=False, policies=opol)
    for a, b, nm in zip(o, g, ("pass", "ref", "off", "mm")):
        bad = np.nonzero(a.astype(np.int64) != b.astype(np.int64))[0]
        assert bad.size == 0, (seed, nm, pols, [(reads[int(i)], int(a[i]), int(b[i])) for i in bad[:3]])
    res.close(); dr.close()
    for d in dev:
        d.close()
    ctx.close()


@pytest.mark.parametrize("seed,n,S", [(1, 70000, 1), (2, 150000, 1), (3, 90000, 3), (4, 1000, 2), (5, 66000, 1)])
def test_random_collapse_matches_counter(seed, n, S):
    """Collapse on random inputs around the partition threshold (65536 reads), all five width classes, reads
    with N, one or several samples: the (sequence -> per-sample count) map and first indices must equal Python's."""
    from collections import Counter
    rng = np.random.default_rng(50 + seed)
    pool = []
    for _ in range(max(n // 6, 10)):
        L = int(rng.choice([16, 18, 20, 22, 22, 24, 27, 30, 31, 32, 33, 40, 64, 65, 100, 128, 129, 150, 200, 255, 256, 300, 1000]))
        pool.append(_rand_seq(rng, L, pn=0.02 if rng.random() < 0.05 else 0.0))
    w = 1.0 / np.arange(1, len(pool) + 1) ** 0.9
    pick = rng.choice(len(pool), size=n, p=w / w.sum())
    reads = [pool[i] for i in pick]
    sid = rng.integers(0, S, size=n).astype(np.int32)
    ctx = _ffi.Context(0)
    raw = _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list(reads))
    uniq = raw.collapse(sid if S > 1 else None, S)
    cnt, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    assert len(set(seqs)) == len(seqs)
    exp = [Counter() for _ in range(S)]
    first_exp = {}
    for i, (r, s) in enumerate(zip(reads, sid if S > 1 else np.zeros(n, np.int32))):
        exp[int(s)][r] += 1
        first_exp.setdefault(r, i)
    assert set(seqs) == set(first_exp)
    for i, r in enumerate(seqs):
        assert [int(x) for x in cnt[i]] == [exp[s].get(r, 0) for s in range(S)], r
        assert int(first[i]) == first_exp[r], r
    uniq.close(); raw.close(); ctx.close()


@pytest.mark.parametrize("seed", range(max(6, N_FUZZ // 4)))
def test_random_text_parses_like_the_host_parser(seed, tmp_path):
    """mirge_reads_parse on random FASTQ / FASTA / line files (random lengths 0-400, N calls, lower case, CRLF or LF,
    with or without a final newline, tiles of the newline scan cut at every offset) against the host parser."""
    from mirge3_amd.collapse import read_fastq_sequences, filter_min_length
    rng = np.random.default_rng(500 + seed)
    ctx = _ffi.Context(0)
    n = int(rng.choice([1, 7, 300, 5000, 40000]))
    fmt = int(rng.integers(1, 4))
    eol = "\r\n" if rng.random() < 0.3 else "\n"
    seqs = []
    for _ in range(n):
        L = int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 64, 65, 128, 255, 256, 257, 400])) if rng.random() < 0.2 else int(rng.integers(14, 60))
        s = _rand_seq(rng, L, pn=0.02 if rng.random() < 0.1 else 0.0)
        if rng.random() < 0.05:
            s = s.lower()
        seqs.append(s)
    if fmt == 3:
        seqs = [s for s in seqs if s] or ["ACGTACGTACGTACGTAC"]
    pad = "x" * int(rng.integers(0, 40))  # shifts every record against the 4 KiB scan tiles
    if fmt == 1:
        text = "".join(f"@r{i} {pad}{eol}{s}{eol}+{eol}{'I' * len(s)}{eol}" for i, s in enumerate(seqs))
    elif fmt == 2:
        text = "".join(f">r{i} {pad}{eol}{s}{eol}" for i, s in enumerate(seqs))
    else:
        text = eol.join(seqs) + eol
    if rng.random() < 0.5 and seqs[-1]:  # (behind an EMPTY last sequence the final line end is what makes it a line at all)
        text = text[:-len(eol)]
    path = tmp_path / "t.txt"
    path.write_text(text, newline="")
    host = read_fastq_sequences(str(path))
    min_len = int(rng.choice([0, 16, 18]))
    exp = filter_min_length(host, min_len)
    dr, n_rec = _ffi.DeviceReads.parse(ctx, path.read_bytes(), fmt, min_len)
    assert n_rec == len(host) and len(dr) == len(exp)
    assert dr.unpack().to_list() == [q.upper() for q in exp.to_list()]
    u = dr.collapse()
    cnt, first = u.counts()
    from collections import Counter
    want = Counter(q.upper() for q in exp.to_list())
    assert dict(zip(u.unpack().to_list(), cnt[:, 0].tolist())) == dict(want)
    u.close(); dr.close(); ctx.close()


def _rseq(rng, L, alphabet="ACGT"):
    return "".join(alphabet[int(c)] for c in rng.integers(0, len(alphabet), size=L))


@pytest.mark.parametrize("seed", range(max(8, N_FUZZ // 2)))
def test_random_trimming_options_equal_the_restated_chain(seed):
    """k_trim under RANDOM options -- none / one / two adapters of 3-40 nt (3' or 5', with an N wildcard), error rate 0-0.3, overlap
    1-8, -n 1-3, --no-indels, read / adapter wildcards, --action none, quality cutoffs at either end, NextSeq trimming, --trim-n,
    -u cuts, the count after every modifier or once, minimum length 0-16 -- on reads that carry whole, damaged, partial or no copies
    of the adapters, N calls, low-quality ends, empty reads and lower-case letters, against the oracle's full-matrix restatement of
    cutadapt's modifiers.  (Written in round 4: the first 60 option sets found that a lower-case g is no dark cycle to NextSeq
    trimming -- cutadapt compares with 'G' -- and that the oracle's 3' search was not case-blind as cutadapt's aligner is.)"""
    rng = np.random.default_rng(7000 + seed)
    ctx = _ffi.Context(0)
    ads = []
    for k in range(int(rng.choice([0, 1, 1, 1, 2]))):
        kind = "front" if rng.random() < 0.35 else "back"
        ad = _rseq(rng, int(rng.integers(3, 41)))
        if kind == "back" and rng.random() < 0.15 and len(ad) > 6:
            p_ = int(rng.integers(1, len(ad) - 1))
            ad = ad[:p_] + "N" + ad[p_ + 1:]
        ads.append((kind, ad))
    opts = {}
    # round 5: anchored adapters (`-g ^A` / `-a B$`) and ONE linked adapter `A...B` (5' part required; under -a anchored with an
    # optional 3' part, under -g regular with a required one; either part may carry an explicit anchor)
    anch = [bool(rng.random() < 0.25) and "N" not in ad for _, ad in ads]
    linked = None
    if len(ads) == 2 and rng.random() < 0.45 and "N" not in ads[0][1]:
        ads = [("front", ads[0][1]), ("back", ads[1][1])]
        under_a = bool(rng.random() < 0.5)
        linked = dict(front=ads[0][1], back=ads[1][1], front_anchored=under_a or anch[0], back_anchored=anch[1] and "N" not in ads[1][1],
                      front_required=True, back_required=not under_a)
    if linked:
        opts["linked"] = linked
    if ads:
        if linked:
            pass
        elif len(ads) == 1 and rng.random() < 0.6:
            opts["adapter"] = ads[0][1]
            if ads[0][0] == "front":
                opts["front"] = True
            if anch[0]:
                opts["anchored"] = True
        else:
            opts["adapters"] = [(k, a, an) for (k, a), an in zip(ads, anch)]
        opts["error_rate"] = float(rng.choice([0.0, 0.05, 0.1, 0.12, 0.2, 0.3]))
        opts["overlap"] = int(rng.integers(1, 9))
        if rng.random() < 0.3:
            opts["times"] = int(rng.integers(2, 4))
        if rng.random() < 0.3:
            opts["indels"] = False
        if rng.random() < 0.2:
            opts["read_wildcards"] = True
        if rng.random() < 0.2:
            opts["adapter_wildcards"] = False
        if rng.random() < 0.1:
            opts["action"] = "none"
    q_back = None if rng.random() < 0.25 else int(rng.integers(0, 31))
    opts["q_back"] = q_back
    if q_back is not None and rng.random() < 0.4:
        opts["q_front"] = int(rng.integers(0, 31))
    if q_back is not None and rng.random() < 0.3:
        opts["nextseq"] = int(rng.integers(5, 31))
    if rng.random() < 0.3:
        opts["trim_n"] = True
    if rng.random() < 0.4:
        cut = [int(c) for c in rng.choice([-5, -2, -1, 1, 2, 4], size=int(rng.integers(1, 3)), replace=False)]
        opts["cut"] = cut[:1] if len(cut) == 2 and (cut[0] > 0) == (cut[1] > 0) else cut
    if not ads and q_back is None and not opts.get("trim_n") and not opts.get("cut"):
        opts["q_back"] = q_back = 10  # a chain without a modifier counts nothing in the reference's worker: the CLI always has -q
    per_mod = bool(rng.random() < 0.5)
    min_len = int(rng.choice([0, 1, 10, 16]))
    recs = []
    for i in range(700):
        s = _rseq(rng, int(rng.integers(0, 45)), "ACGTN" if rng.random() < 0.1 else "ACGT")
        for kind, ad in ads:
            x = list(ad.replace("N", "ACGT"[int(rng.integers(0, 4))]))
            r = rng.random()
            if r < 0.2 and len(x) > 1:
                x[int(rng.integers(0, len(x)))] = "ACGT"[int(rng.integers(0, 4))]
            elif r < 0.3 and len(x) > 3:
                del x[int(rng.integers(1, len(x) - 1))]
            elif r < 0.4 and len(x) > 2:
                x.insert(int(rng.integers(1, len(x) - 1)), "ACGT"[int(rng.integers(0, 4))])
            elif r < 0.5:
                x = x[:int(rng.integers(1, len(x) + 1))] if kind == "back" else x[-int(rng.integers(1, len(x) + 1)):]
            elif r < 0.6:
                x = []
            x = "".join(x)
            if rng.random() < 0.1 and x:
                x = x[:len(x) // 2] + "N" + x[len(x) // 2 + 1:]
            s = (x + s) if kind == "front" else (s + x + _rseq(rng, int(rng.integers(0, 8))))
        if rng.random() < 0.03:
            s = ""
        if rng.random() < 0.05:
            s = s.lower()
        elif rng.random() < 0.03:
            s = "".join(c.lower() if rng.random() < 0.3 else c for c in s)
        s = s[:120]
        q = rng.integers(33, 74, size=len(s)).astype(np.uint8)
        if rng.random() < 0.3 and len(s):
            q[-int(rng.integers(1, len(s) + 1)):] = rng.integers(33, 45, size=1)[0]
        if rng.random() < 0.1 and len(s):
            q[:int(rng.integers(1, 6))] = 35
        recs.append((s, q.tobytes().decode()))
    text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
    a1 = ads[0] if ads else (None, None)
    a2 = ads[1] if len(ads) > 1 else (None, None)
    if any(k == "front" and "N" in a for k, a in ads):
        pytest.skip("N in a 5' adapter is refused by the C ABI")
    lkw = dict(anchored=bool(ads) and anch[0], anchored2=len(ads) > 1 and anch[1])
    if linked:
        lkw = dict(anchored=linked["front_anchored"], anchored2=linked["back_anchored"], linked=True, front_required=True,
                   back_required=linked["back_required"])
    trim = _ffi.MirgeTrim.make(adapter=a1[1], front=a1[0] == "front", adapter2=a2[1], front2=a2[0] == "front", **lkw,
                               quality_back=-1 if q_back is None else q_back, quality_front=opts.get("q_front", 0), nextseq=opts.get("nextseq", -1),
                               min_overlap=opts.get("overlap", 3), error_rate=opts.get("error_rate", 0.12), trim_n=opts.get("trim_n", False),
                               cut=opts.get("cut", []), count_per_modifier=per_mod, times=opts.get("times", 1), indels=opts.get("indels", True),
                               read_wildcards=opts.get("read_wildcards", False), adapter_wildcards=opts.get("adapter_wildcards", True),
                               action=opts.get("action", "trim"))
    raw, n_rec = _ffi.DeviceReads.parse(ctx, text, 1, min_len, trim)
    uniq = raw.collapse()
    cnt, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    order = np.argsort(first, kind="stable")
    got = [(seqs[i], int(cnt[i, 0])) for i in order]
    want = {}
    for k, v in oracle.trimmed_counts(recs, dict(opts), min_len, per_mod).items():  # the oracle keeps a read's case, the device packs letters
        want[k.upper()] = want.get(k.upper(), 0) + v
    assert n_rec == len(recs) and got == list(want.items()), opts
    uniq.close(); raw.close(); ctx.close()


@pytest.mark.parametrize("seed", range(max(6, N_FUZZ // 3)))
def test_random_umi_options_equal_the_restated_worker(seed, tmp_path):
    """mirge_reads_parse_umi under RANDOM options -- -umi f,b with f 0-8 and b 0-12, --qiagenumi, -udd, with or without the 3'
    adapter, quality cutoffs, the count after every modifier or once, minimum length 0-16 -- on UMI-library reads (damaged or cut
    adapters, reads that end inside the UMI, empty reads, low-quality ends; lower-case reads only without -udd: the device packs
    letters, so 'acgt' and 'ACGT' are one molecule here and two dictionary keys in the reference) against the oracle's restatement
    of the worker's UMI branches and baking's UMI stage: the dictionary in order, 'Trimmed Reads (all)', <sample>_umiCounts.csv."""
    from mirge3_amd.collapse import parse_sample
    rng = np.random.default_rng(9000 + seed)
    ctx = _ffi.Context(0)
    qiagen = bool(rng.random() < 0.35)
    f = 0 if qiagen and rng.random() < 0.8 else int(rng.integers(0, 9))
    b = int(rng.integers(0, 13))
    dedup = bool(rng.random() < 0.5)
    adapter = _rseq(rng, int(rng.integers(8, 30)))
    q_back, q_front = int(rng.integers(0, 25)), int(rng.choice([0, 0, 8, 15]))
    per_mod = bool(rng.random() < 0.5)
    min_len = int(rng.choice([0, 1, 10, 16]))
    use_adapter = qiagen or rng.random() < 0.8
    inserts = [_rseq(rng, int(rng.integers(0, 40))) for _ in range(40)] + ["A" * 20, "ACGTACGTACGTACGTACGGACGTACGTACGTACGTAC"]
    umis = [_rseq(rng, f + b) for _ in range(7)]
    recs = []
    for i in range(1200):
        ins = inserts[int(rng.integers(0, len(inserts)))]
        u = umis[int(rng.integers(0, len(umis)))]
        ad = list(adapter)
        kind = int(rng.integers(0, 9))
        if kind == 1:
            ad[int(rng.integers(0, len(ad)))] = "ACGT"[int(rng.integers(0, 4))]
        elif kind == 2 and len(ad) > 3:
            del ad[int(rng.integers(1, len(ad) - 1))]
        ad = "".join(ad)
        ext = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC"
        seq = (ins + ad + u[f:] + ext) if qiagen else (u[:f] + ins + u[f:] + ad + ext)
        seq = seq[:int(rng.integers(10, 100))] if kind in (3, 4) else seq[:90]
        if kind == 5:
            seq = ins
        if kind == 6:
            seq = ""
        if kind == 7 and rng.random() < 0.5 and not dedup:
            seq = seq.lower()
        q = np.full(len(seq), ord("I"), dtype=np.uint8)
        if rng.random() < 0.25 and len(seq):
            q[-int(rng.integers(1, 12)):] = rng.integers(33, 48, size=1)[0]
        if rng.random() < 0.05 and len(seq):
            q[:int(rng.integers(1, 4))] = 35
        recs.append((seq, q.tobytes().decode()))
    text = "".join(f"@r{i}\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(recs)).encode()
    trim = _ffi.MirgeTrim.make(adapter=adapter if use_adapter else None, quality_back=q_back, quality_front=q_front, count_per_modifier=per_mod)
    raw, n_rec = parse_sample(ctx, text, min_len, trim, _ffi.MirgeUmi.make(f, b, qiagen=qiagen, dedup=dedup), tmp_path, "S")
    uniq = raw.collapse()
    cnt, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    order = np.argsort(first, kind="stable")
    got, n = [(seqs[i], int(cnt[i, 0])) for i in order], len(raw)
    uniq.close(); raw.close(); ctx.close()
    csvf = tmp_path / "S_umiCounts.csv"
    csv = csvf.read_text() if csvf.exists() else None
    o = dict(q_back=q_back, q_front=q_front)
    if use_adapter:
        o["adapter"] = adapter
    keys = oracle.umi_worker_reads(recs, o, f, b, min_len, qiagen, per_mod)
    want, trimmed, rows = oracle.umi_baking(keys, f, b, min_len, dedup)
    wu = {}
    for k, v in want:
        wu[k.upper()] = wu.get(k.upper(), 0) + v
    assert n_rec == len(recs) and n == trimmed and got == list(wu.items()), (f, b, qiagen, dedup, per_mod, min_len, use_adapter)
    assert (csv is None) == (rows is None) and (csv is None or csv == "".join(rows))


@pytest.fixture(scope="module")
def _fz():
    from mirge3_amd import synth
    from mirge3_amd.cascade import Cascade
    ctx = _ffi.Context(0)
    sl = synth.make_libraries(seed=77, scale="ci")
    return ctx, sl, Cascade(ctx, sl.libs)


def _fz_reads(rng, sl, it, n):
    from mirge3_amd import synth
    reads = synth.make_reads(sl, n, seed=2000 + it, n_frac=0.03 if it % 3 == 0 else 0.0)
    if it % 4 == 1:  # wide and long reads, runs of one letter, N runs
        extra = FlatSeqs.from_list([("ACGTN"[int(rng.integers(0, 5))]) * int(rng.integers(1, 700)) for _ in range(40)] + ["ACGT" * 70])
        reads = FlatSeqs(np.concatenate([reads.data, extra.data]), np.concatenate([reads.offsets, extra.offsets[1:] + reads.offsets[-1]]))
    return reads


@pytest.mark.parametrize("seed", range(max(6, N_FUZZ // 3)))
def test_random_count_join_and_per_read_csv(seed, _fz, tmp_path):
    """mirge_count_join (both its forms) against numpy sums, and mapped.csv / unmapped.csv formatted on the GPU against the host
    formatter and -- small cases -- against plain Python, for random sample counts (1-16), read sets of 1 to 400 k reads with N
    calls, wide and long reads, count matrices with zeros and cells up to 2^32 - 1, either row order, with and without the
    spike-in column."""
    from mirge3_amd.fastpath import names_by_pass, PASS_COLUMNS
    ctx, sl, casc = _fz
    rng = np.random.default_rng(15000 + seed)
    n = int(rng.choice([1, 40, 3000, 80000, 400000]))
    S = int(rng.choice([1, 1, 2, 3, 7, 16]))
    reads = _fz_reads(rng, sl, seed, n)
    n = len(reads)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq = raw.collapse(rng.integers(0, S, size=n).astype(np.int32), S) if S > 1 else raw.collapse()
    if seed % 3 == 2:
        w = rng.integers(0, 1 << 32, size=uniq.counts()[0].shape, dtype=np.int64).astype(np.uint32)
        w[rng.random(w.shape) < 0.3] = 0
        uniq.set_counts(w)
    res = casc.run(uniq)
    counts, first = uniq.counts()
    ps, ref, off, mm = res.fetch()
    nm = len(sl.libs["mirna"])
    c = counts.astype(np.int64)
    we, wi = np.zeros((nm, S), np.int64), np.zeros((nm, S), np.int64)
    np.add.at(we, ref[ps == 0], c[ps == 0])
    np.add.at(wi, ref[ps == 8], c[ps == 8])
    for mode in ("1", "0"):
        os.environ["MIRGE_JOIN_ROWS"] = mode
        try:
            cls, ex, iso = _ffi.count_join(ctx, uniq, res, 0, 8, nm)
        finally:
            os.environ.pop("MIRGE_JOIN_ROWS")
        assert all(np.array_equal(cls[p], c[ps == p].sum(axis=0)) for p in range(res.n_pass)) and np.array_equal(ex, we) and np.array_equal(iso, wi)
    order = uniq.first_appearance_order() if (S == 1 and seed % 2 == 0) else uniq.sorted_order()
    n_cols = 10 if seed % 5 == 0 else 9
    header = ",".join(["Sequence", "annotFlag"] + PASS_COLUMNS[:n_cols] + [f"S{k}" for k in range(S)]) + "\n"
    nb = names_by_pass(casc)
    dev, host = tmp_path / "dev", tmp_path / "host"
    dev.mkdir(); host.mkdir()
    assert _ffi.annotation_csv_device(ctx, uniq, res, dev / "mapped.csv", dev / "unmapped.csv", header, order, list(range(casc.n_pass)), n_cols, nb)
    seqs = uniq.unpack()
    _ffi.annotation_csv(host / "mapped.csv", host / "unmapped.csv", header, seqs, ps, ref, counts, order, list(range(casc.n_pass)), n_cols, nb)
    for f in ("mapped.csv", "unmapped.csv"):
        assert (dev / f).read_bytes() == (host / f).read_bytes(), f
    if n <= 3000:
        sl_, lines = seqs.to_list(), {True: [header], False: [header]}
        for i in order:
            names = [""] * n_cols
            if 0 <= ps[i] < n_cols:
                names[ps[i]] = nb[ps[i]].get(int(ref[i]))
            lines[bool(ps[i] >= 0)].append(",".join([sl_[i], "1" if ps[i] >= 0 else "0"] + names + [str(int(x)) for x in counts[i]]) + "\n")
        assert (host / "mapped.csv").read_text() == "".join(lines[True]) and (host / "unmapped.csv").read_text() == "".join(lines[False])
    res.close(); uniq.close(); raw.close()


@pytest.mark.parametrize("seed", range(max(5, N_FUZZ // 4)))
def test_random_variant_tally_matches_the_restatement(seed):
    """mirge_variant_tally (k_member_list + k_tally: the A-to-I report's counting) under random libraries, read mixes (substitutions
    at every position incl. N, 5' / 3' shifts of -3 .. +3 taken from the hairpin, non-templated ends), 1-3 samples, merged families or
    one family per miRNA, a retained mask or none, RPM gates from 0 to "everything is a member": every per-read and per-family
    field and the 12-way census against the string restatement the reference's own functions pinned."""
    from mirge3_amd import a2i, synth
    from mirge3_amd.cascade import Cascade
    rng = np.random.default_rng(21000 + seed)
    ctx = _ffi.Context(0)
    sl = synth.make_libraries(seed=100 + seed % 5, scale="ci")
    mir, hp = sl.libs["mirna"].seqs.to_list(), sl.libs["hairpin"].seqs.to_list()
    reads = synth.make_reads(sl, int(rng.choice([300, 5000, 20000])), seed=seed, mix=dict(exact=0.3, isomir=0.6, random=0.1),
                             n_frac=0.02 if seed % 3 == 0 else 0).to_list()
    for i in range(0, len(mir), int(rng.integers(2, 6))):
        s_ = mir[i]
        for q, b in enumerate(s_):
            if rng.random() < 0.15:
                reads += [s_[:q] + "ACGTN"[int(rng.integers(0, 5))] + s_[q + 1:]] * int(rng.integers(1, 6))
        h, o = int(sl.mir_hairpin[i]), int(sl.mir_hairpin_off[i])
        for d5 in (-2, -1, 0, 1, 2):
            for d3 in (-3, -1, 0, 1, 2, 3):
                if rng.random() < 0.15:
                    reads.append(hp[h][max(o + d5, 0):o + len(s_) + d3])
        reads += [s_ + "A", s_ + "TT", "G" + s_, s_[:-1] + "N", s_[1:] + "C"]
    reads = [r for r in reads if len(r) >= 16]
    S = int(rng.choice([1, 2, 3]))
    casc = Cascade(ctx, sl.libs)
    raw = _ffi.DeviceReads.pack(ctx, FlatSeqs.from_list(reads))
    uniq = raw.collapse(rng.integers(0, S, size=len(reads)).astype(np.int32), S) if S > 1 else raw.collapse()
    res = casc.run(uniq)
    ps, ref, off, mm = res.fetch()
    counts, _ = uniq.counts()
    useq = uniq.unpack().to_list()
    if seed % 2 == 0:
        fam_names, fam_of_ref = a2i.families(sl.libs["mirna"].names, sl.merges)
        first_member = {}
        for r, f in enumerate(fam_of_ref):
            first_member.setdefault(int(f), r)
        targets = [mir[first_member[f]] for f in range(len(fam_names))]
    else:
        fam_of_ref, targets = np.arange(len(mir)), mir
    retained = (rng.random(len(useq)) < rng.choice([0.5, 0.8, 1.0])).astype(np.uint8) if seed % 3 else None
    freq = np.array([float(rng.choice([0.0, 0.4, 1.5, 1e300])) for _ in range(S)])
    g = a2i.tally(casc, uniq, res, fam_of_ref, FlatSeqs.from_list(targets), retained, freq, per_read=True)
    o = oracle.variant_tally(useq, counts.astype(np.int64), ps, ref, fam_of_ref, targets, retained, freq)
    for k in ("diag", "state", "n_seqs", "seq_true", "count_true", "canon", "kept_exact", "census"):
        assert np.array_equal(g[k], o[k]), k
    res.close(); uniq.close(); raw.close(); casc.close(); ctx.close()

