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
    deps = [SRC] + [os.path.join(HERE, "..", "mirge3.0_amd", "csrc", f) for f in ("mirge_core.hpp", "mirge_libbuild.hpp")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-o", SO, SRC])
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
    allr = FlatSeqs.from_list(reads.to_list() + extra)
    o = oracle.cascade(allr.data, allr.offsets, oracle_libs_from(sl.libs), n_pass=9, indexed=True)
    h = hostsim_cascade(allr, sl.libs, 9)
    for a, b, nm in zip(o, h, ("pass", "ref", "off", "mm")):
        bad = np.nonzero(a.astype(np.int64) != b.astype(np.int64))[0]
        assert bad.size == 0, (nm, bad[:5], [allr.get(int(i)) for i in bad[:3]], a[bad[:5]], b[bad[:5]])
    assert (o[0] >= 0).mean() > 0.5
