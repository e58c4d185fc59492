"""Length-binned unique reads: what does a wave that holds ONE read length buy the cascade?  (round 3, review item 4)

The unique reads of the bench's 10 M-read sample, packed (a) in the order the collapse emitted them, (b) in a random
order, (c) grouped by length (stable inside a length), each through mirge_cascade_run with every launch bracketed by
HIP events: per-pass kernel times of the SAME build on the three orders.  (c) is the upper bound of what emitting the
unique reads length-binned can give the current kernels; a build with wave-uniform (scalar) plans is compared with
tools/ab_so.py on top.

  python tools/len_bin_experiment.py [reads] [rounds] [only: collapse|random|grouped]
(profiles/collect_lenbin.sh runs one order per rocprofv3 --pmc pass and compares the counters of the isomiR pass.)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mirge3_amd  # noqa: E402,F401
from mirge3_amd import _ffi, synth  # noqa: E402
from mirge3_amd.cascade import Cascade  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    sl = synth.make_libraries(seed=20260101, scale="full")
    ctx = _ffi.Context(0)
    casc = Cascade(ctx, sl.libs)
    reads = synth.make_reads_chunked(sl, n, seed=1000)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq = raw.collapse()
    useq = uniq.unpack()  # handle order = the order the collapse emitted them
    uniq.close(); raw.close()
    rng = np.random.default_rng(1)
    orders = {
        "collapse order": np.arange(len(useq)),
        "random order": rng.permutation(len(useq)),
        "grouped by length": np.argsort(useq.lengths, kind="stable"),
    }
    # grouped by length, but a workgroup's contiguous segment (k_pass gives every workgroup one) made of 256-read tiles
    # dealt round-robin from the sorted order: every workgroup sees every length (balance), waves stay mostly one length
    G = 6 * 256
    srt = orders["grouped by length"]
    n_u = len(srt)
    T = -(-(-(-n_u // G)) // 256)  # tiles per workgroup segment
    tile_of = np.arange(n_u) // 256
    seg, k = tile_of % G, tile_of // G
    dest = (seg * T + k) * 256 + np.arange(n_u) % 256
    orders["grouped, tiles dealt"] = srt[np.argsort(dest, kind="stable")]
    only = sys.argv[3] if len(sys.argv) > 3 else None
    if only:
        orders = {k: v for k, v in orders.items() if k.startswith(only)}
    ref_ann = None
    for name, o in orders.items():
        dr = _ffi.DeviceReads.pack(ctx, useq.take(o))
        res = casc.run(dr)  # warm: tables, plans
        ann = res.fetch()
        res.close()
        inv = np.empty_like(o)
        inv[o] = np.arange(len(o))
        if ref_ann is None:
            ref_ann = ann
        else:  # same answers whatever the order
            assert all(np.array_equal(a[inv], b) for a, b in zip(ann, ref_ann)), name
        ctx.profile(True); ctx.profile_only(""); ctx.profile_reset()
        t = time.perf_counter()
        for _ in range(rounds):
            res = casc.run(dr)
            res.close()
        ctx.sync()
        wall = (time.perf_counter() - t) / rounds * 1e3
        recs = {nm: (ms / l, l // rounds) for nm, l, ms, u in ctx.profile_records() if l and nm.startswith(("k_pass", "k_resolve", "k_cascade"))}
        ctx.profile(False)
        tot = sum(ms * k for ms, k in recs.values())
        print(f"{name:20s} wall {wall:7.3f} ms  kernels {tot:7.3f} ms  " + "  ".join(f"{nm}={ms:.3f}" for nm, (ms, k) in sorted(recs.items()) if nm.endswith(".w1")))
        dr.close()


if __name__ == "__main__":
    main()
