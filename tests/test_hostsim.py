"""Device arithmetic checked on the CPU: the header the kernels are compiled from
(mirge_core.hpp + mirge_libbuild.hpp) driven by tests/hostsim/hostsim.cpp, against the oracle.

This is a debugging aid for a container without a GPU, not a product path (see hostsim.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle
from helpers import CASES, GoldenCase, PASS_LIBKEY, oracle_libs_from
import mirge3_amd  # noqa: F401
from mirge3_amd import synth
from mirge3_amd.cascade import policies
from mirge3_amd.seqio import FlatSeqs

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostsim", "hostsim.cpp")
SO = os.path.join(HERE, "hostsim", "_build", "libhostsim.so")


def _sim():
    deps = [SRC] + [os.path.join(HERE, "..", "mirge3.0_amd", "csrc", f) for f in ("mirge_core.hpp", "mirge_libbuild.hpp", "mirge_isotype.hpp")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-pthread", "-o", SO, SRC])
    return C.CDLL(SO)


def hostsim_cascade(reads: FlatSeqs, libs, n_pass):
    so = _sim()
    n = len(reads)
    keep = []
    seqp = (C.c_char_p * n_pass)()
    offp = (C.c_void_p * n_pass)()
    nref = (C.c_int64 * n_pass)()
    for p in range(n_pass):
        lib = libs[PASS_LIBKEY[p]]
        d = np.ascontiguousarray(lib.seqs.data)
        o = np.ascontiguousarray(lib.seqs.offsets)
        keep += [d, o]
        seqp[p] = C.cast(d.ctypes.data, C.c_char_p)
        offp[p] = o.ctypes.data
        nref[p] = len(lib)
    pol = policies(n_pass)
    polarr = (type(pol[0]) * n_pass)(*pol)
    ps = np.empty(n, dtype=np.int8); ref = np.empty(n, dtype=np.int32)
    off = np.empty(n, dtype=np.int32); mm = np.empty(n, dtype=np.int8)
    rd = np.ascontiguousarray(reads.data); ro = np.ascontiguousarray(reads.offsets)
    rc = so.hostsim_cascade(C.c_void_p(rd.ctypes.data), C.c_void_p(ro.ctypes.data), C.c_int64(n), seqp,
                            C.cast(offp, C.POINTER(C.c_void_p)), nref, polarr, C.c_int32(n_pass),
                            C.c_void_p(ps.ctypes.data), C.c_void_p(ref.ctypes.data),
                            C.c_void_p(off.ctypes.data), C.c_void_p(mm.ctypes.data))
    assert rc == 0
    return ps, ref, off, mm


@pytest.mark.parametrize("name", CASES)
def test_hostsim_golden(name):
    case = GoldenCase(name)
    ps, ref, off, mm = hostsim_cascade(case.reads, case.libs, case.n_pass)
    exp = case.expected_annotation()
    for i, s in enumerate(case.seqs):
        p = int(ps[i])
        nm = case.lib_of_pass(p).names[int(ref[i])] if p >= 0 else ""
        assert (p, nm) == exp[s], s


def test_hostsim_vs_oracle_ci_scale():
    sl = synth.make_libraries(seed=77, scale="ci")
    reads = synth.make_reads(sl, 30000, seed=5, n_frac=0.01)
    # long reads (W=2 and W=4 groups) from the long-RNA libraries
    extra = []
    rng = np.random.default_rng(3)
    for key in ("mrna", "ncrna_others", "rrna"):
        lib = sl.libs[key]
        for _ in range(60):
            r = int(rng.integers(0, len(lib)))
            s = lib.seqs.get(r)
            L = int(rng.integers(51, 128))
            if len(s) <= L:
                continue
            a = int(rng.integers(0, len(s) - L))
            x = list(s[a:a + L])
            for _ in range(int(rng.integers(0, 3))):
                q = int(rng.integers(0, L)); x[q] = "ACGT"[("ACGT".index(x[q]) + 1) % 4] if x[q] in "ACGT" else "A"
            extra.append("".join(x))
    # very short reads (below the default --minimum-length): 1..15 nt substrings, some mutated,
    # T-tailed heads of every length, poly-runs
    short = []
    for key in ("mirna", "pre_trna", "mature_trna", "snorna", "mrna"):
        lib = sl.libs[key]
        for L in range(1, 16):
            for _ in range(6):
                s = lib.seqs.get(int(rng.integers(0, len(lib))))
                a = int(rng.integers(0, len(s) - L))
                x = list(s[a:a + L])
                if rng.random() < 0.4 and x[0] in "ACGT":
                    q = int(rng.integers(0, L)); x[q] = "ACGT"[("ACGT".index(x[q]) + 1) % 4] if x[q] in "ACGT" else "A"
                short.append("".join(x))
                short.append("".join(x) + "T" * int(rng.integers(3, 7)))
    short += ["A", "T", "TTT", "TTTT", "ATTT", "N", "NNNN", "ACGTTTT", "G" * 15, "AC" * 7]
    allr = FlatSeqs.from_list(reads.to_list() + extra + short)
    o = oracle.cascade(allr.data, allr.offsets, oracle_libs_from(sl.libs), n_pass=9, indexed=True)
    h = hostsim_cascade(allr, sl.libs, 9)
    for a, b, nm in zip(o, h, ("pass", "ref", "off", "mm")):
        bad = np.nonzero(a.astype(np.int64) != b.astype(np.int64))[0]
        assert bad.size == 0, (nm, bad[:5], [allr.get(int(i)) for i in bad[:3]], a[bad[:5]], b[bad[:5]])
    assert (o[0] >= 0).mean() > 0.5


def test_probe_plan_covers_every_mismatch_placement():
    """Exhaustive proof of the filter: for every probe family (plain / recursive / pair-of-blocks), every seed
    length, probe length K and every placement of <= mm mismatches inside the seed, some probe's exact blocks
    avoid all of them (so the true window is among its candidates), blocks stay inside the seed, and shapes fit
    the registry.  Also for the family the cost model picks at several library sizes."""
    import itertools
    so = _sim()
    out = (C.c_int8 * 36)()
    picked = set()
    for mode, mm in ((0, 0), (0, 1), (1, 1), (1, 2), (1, 3), (0, 2)):
        for K in range(8, 15):
            for L in range(mm + 1, 50):
                S = min(L, 28) if mode == 0 else L
                plans = [(-1, npos) for npos in (2000, 60000, 11_000_000, 140_000_000)]
                plans.append((0, 1))
                if mm in (1, 2) and S >= 2 * (mm + 1):
                    plans.append((1, 1))
                if mm in (1, 2) and S >= mm + 2:
                    plans.append((2, 1))
                seen = set()
                for scheme, npos in plans:
                    n = so.hostsim_probe_plan(mode, mm, 28, L, K, C.c_int64(npos), scheme, out)
                    if scheme < 0:
                        picked.add(so.hostsim_plan_scheme(mode, mm, 28, L, K, C.c_int64(npos)))
                    sig = bytes(out[:4 * n])
                    if sig in seen:
                        continue
                    seen.add(sig)
                    probes = []
                    assert 1 <= n <= 9
                    for q in range(n):
                        a1, k1, gap, k2 = out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]
                        if k1 <= 0:
                            continue
                        cover = set(range(a1, a1 + k1)) | set(range(a1 + k1 + gap, a1 + k1 + gap + k2))
                        assert min(cover) >= 0 and max(cover) < S, (mode, mm, K, L, scheme, q)
                        # an mm = 0 policy may take k = 15 in a library that has outgrown K = 14 (one plain table)
                        assert k1 + k2 <= (15 if (mm == 0 and K == 14 and k2 == 0) else K) and 0 <= gap < 32 and (k2 > 0 or gap == 0)
                        probes.append(cover)
                    budget = mm  # mismatches allowed inside the seed region
                    for r in range(budget + 1):
                        if S > 34 and r == 2:
                            combos = itertools.islice(itertools.combinations(range(S), r), 0, None, 7)
                        else:
                            combos = itertools.combinations(range(S), r)
                        for bad in combos:
                            bs = set(bad)
                            assert any(not (c & bs) for c in probes), (mode, mm, K, L, scheme, bad)
    assert picked == {0, 1, 2}  # the cost model uses every family somewhere


def test_core_arithmetic_is_asan_ubsan_clean(tmp_path):
    """mirge_core.hpp / mirge_libbuild.hpp / mirge_isotype.hpp under AddressSanitizer + UBSan on the CPU build (the GPU
    pool has no device sanitizer): random libraries with reference Ns, reads of 1..128 nt with Ns,
    T tails and mismatches, all nine policies; the isomiR typing's two forms on 150 k pairs."""
    exe = str(tmp_path / "sanitize")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17",
                           "-Wno-unknown-pragmas", os.path.join(HERE, "hostsim", "sanitize_main.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rc=0" in r.stdout and "annotated=" in r.stdout
    assert "differing=0" in r.stdout  # mirge_isotype.hpp, both forms of the isomiR typing, 150 k pairs under the sanitizers


def test_pure_host_functions_are_asan_ubsan_clean(tmp_path):
    """native_host.hpp -- the mapped.csv / unmapped.csv formatter (mirge_annotation_csv), the GFF3 writer, the text of a merged
    library, the argument checks of mirge_lib_create_packed: the translation unit the product compiles -- under AddressSanitizer +
    UBSan with g++: names that need CSV quoting, empty libraries and references, reads of 0 / 255 / 600 nt, 1 and 17 samples,
    indices that must be refused, records at the text limit; the formatter's files are compared with a std::string restatement.
    Also native_gz.hpp (mirge_gz_inflate): gzip members of FASTQ-like text at levels 1 / 6 / 9, with sync flushes, on 1-8 threads,
    BGZF files, and what it must refuse -- a flipped bit, a truncated file, a second member, too little room."""
    exe = str(tmp_path / "host_only")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-pthread",
                           "-Wno-unknown-pragmas", os.path.join(HERE, "hostsim", "host_only_main.cpp"), "-o", exe, "-lz"])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "host-only functions clean" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    assert "inflated in parallel and equal" in r.stdout  # native_gz.hpp: the block-start search and the history resolution ran


def test_gz_inflater_survives_damaged_files(tmp_path):
    """tests/hostsim/gz_fuzz_main.cpp under ASan/UBSan: .gz samples (one member, pigz-style flushes, two members, BGZF) with flipped
    bits, overwritten / zeroed / repeated spans and cut ends, on 1-6 threads.  No sanitizer report, and whatever mirge_gz_inflate
    accepts is byte for byte what zlib makes of the same bytes (4 400 mutants of this driver ran clean when it was written, also under
    TSan; the suite runs 120)."""
    exe = str(tmp_path / "gz_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-pthread",
                           "-Wno-unknown-pragmas", os.path.join(HERE, "hostsim", "gz_fuzz_main.cpp"), "-o", exe, "-lz"])
    r = subprocess.run([exe, "120", "11"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "gz fuzz clean" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    assert " 0 inflated in parallel" not in r.stdout  # the undamaged mutants took the parallel route


def hostsim_isotype(master: str, read: str, precursor: str, form: str = "arrays"):
    """``form``: "arrays" = mirge_isotype (per-thread arrays, the general path), "registers" = mirge_isotype_fast (bit planes;
    what k_isotype runs for every pair it can take) -- None when the pair is not one for that form"""
    so = _sim()
    kind, start, end = C.c_int32(), C.c_int32(), C.c_int32()
    var, cig = C.create_string_buffer(400), C.create_string_buffer(400)
    start0 = (precursor.find(master) + 1) if precursor != "" else 1
    fn = so.hostsim_isotype if form == "arrays" else so.hostsim_isotype_fast
    rc = fn(master.encode(), len(master), read.encode(), len(read), precursor.encode(), len(precursor), start0,
            C.byref(kind), C.byref(start), C.byref(end), var, cig)
    if form == "registers" and rc == -2:
        return None
    assert rc == 0
    return ({0: "none", 1: "ref_miRNA", 2: "isomiR"}[kind.value], start.value, end.value, var.value.decode(), cig.value.decode())


@pytest.mark.parametrize("form", ["arrays", "registers"])
@pytest.mark.parametrize("case_name", ["case4_gff_a2i", "case6_gff_a2i"])
def test_isotype_core_equals_the_oracle_on_the_reference_lines(case_name, form):
    """every line of the GFF the reference wrote (golden cases 4 and 6), through the header k_isotype is compiled from -- both
    forms of the typing: the one on per-thread arrays and the register-resident one the kernel prefers"""
    from test_a2i_gff_oracle import _case4_gff_tables
    case, mat, pre, pre_of = _case4_gff_tables(case_name)
    n = 0
    for ln in open(os.path.join(case.dir, "sample_miRge3.gff")):
        if ln.startswith("#"):
            continue
        f = ln.rstrip("\n").split("\t")
        attrs = dict(x.split("=", 1) for x in f[8].split("; "))
        got = hostsim_isotype(mat[f[0]], attrs["Read"], pre[pre_of[f[0]]], form)
        assert got == (f[2], int(f[3]), int(f[4]), attrs["Variant"], attrs["Cigar"]), (f[0], mat[f[0]], attrs["Read"])
        n += 1
    assert n > 650


def test_isotype_core_equals_the_oracle_on_random_pairs():
    """mutated, shifted, extended and unrelated reads against random canonicals inside random precursors (canonical
    at the very start / end of the precursor, absent from it, empty precursor): the difflib restatement and the two
    list-rewriting passes are compared on every field"""
    rng = np.random.default_rng(11)
    n = 0
    shapes = set()
    for it in range(6000):
        la = int(rng.integers(16, 27))
        master = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=la))
        if it % 9 == 0:
            master = master[:6] + master[5] * 4 + master[10:]  # homopolymer runs: ambiguous diffs
        mode = it % 7
        lead = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(0, 12) if mode != 1 else 0)))
        trail = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(0, 12) if mode != 2 else 0)))
        precursor = "" if mode == 3 else (lead + trail if mode == 4 else lead + master + trail)
        d5, d3 = int(rng.integers(-3, 4)), int(rng.integers(-4, 5))
        body = master[max(d5, 0):la + min(d3, 0)]
        src = lead + master + trail
        o = len(lead)
        five = src[max(o + min(d5, 0), 0):o] if rng.random() < 0.6 else "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=max(-d5, 0)))
        three = src[o + la:o + la + max(d3, 0)] if rng.random() < 0.6 else "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=max(d3, 0)))
        read = list(five + body + three)
        for _ in range(int(rng.choice([0, 0, 1, 1, 2, 3]))):
            if read:
                read[int(rng.integers(0, len(read)))] = "ACGTN"[int(rng.integers(0, 5))]
        if it % 50 == 0:
            read = list("".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(14, 31)))))
        read = "".join(read)
        if len(read) < 10:
            continue
        want = oracle.gff_record(master, read, precursor)
        got = hostsim_isotype(master, read, precursor)
        assert got == want, (master, read, precursor, got, want)
        fast = hostsim_isotype(master, read, precursor, "registers")
        assert fast == want, (master, read, precursor, fast, want)  # (every pair here is one the register form takes)
        shapes.add(want[3].split(":")[0].split(",")[0])
        n += 1
    assert n > 5000 and len(shapes) >= 8


def test_isotype_register_form_equals_the_array_form_on_millions_of_pairs():
    """k_isotype's register-resident typing (mirge_isotype_fast: sequences and the aligned lists as bit planes, difflib's longest
    match along diagonals) against the array form it replaced on the hot path, field for field, on 3 M pairs made in C++: shifted,
    extended (templated or not), substituted, N-called, single-base indel, homopolymer, tandem-repeat and unrelated reads; the
    canonical at the start / end of the precursor, absent from it, empty precursor."""
    so = _sim()
    so.hostsim_isotype_fuzz.restype = C.c_int64
    so.hostsim_isotype_fuzz.argtypes = [C.c_uint64, C.c_int64, C.POINTER(C.c_int64), C.c_char_p, C.c_char_p, C.c_char_p]
    bad = C.c_int64()
    a, b, p = C.create_string_buffer(100), C.create_string_buffer(200), C.create_string_buffer(300)
    for seed in (101, 202, 303):
        took = so.hostsim_isotype_fuzz(seed, 1_000_000, C.byref(bad), a, b, p)
        assert bad.value == 0, (seed, bad.value, a.value, b.value, p.value)
        assert took > 900_000  # (the rest: more than 64 columns)

