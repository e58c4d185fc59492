"""Would bucketing the survivors by the top bits of their probe key pay (review round 3, item 3b: 7.5 M translation misses,
DRAM page locality)?  The upper bound, without writing the kernel: the unique reads of the bench's 10 M-read sample re-packed on
the host in the order of their first probe key of the merged snoRNA / rRNA / ncRNA pass (bases 0..12, the table's own address
order) and of the mRNA pass (bases 0..14) -- every workgroup's segment is then a narrow address range of that table -- through the
SAME build, one launch per pass (MIRGE_BULK_FUSED=0), every launch bracketed by HIP events.

  MIRGE_BULK_FUSED=0 python tools/key_sort_experiment.py [reads] [rounds]
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


def kmer_key(useq, k):
    """little-endian 2-bit integer of bases 0..k-1 (the probe tables' key); reads shorter than k or with an N there sort last"""
    L = useq.lengths
    code = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    key = np.zeros(len(useq), dtype=np.uint64)
    bad = L < k
    starts = useq.offsets[:-1]
    for j in range(k):
        idx = np.minimum(starts + j, useq.data.shape[0] - 1)
        c = code[useq.data[idx]]
        bad |= c == 255
        key |= (c.astype(np.uint64) & np.uint64(3)) << np.uint64(2 * j)
    key[bad] = np.uint64(1) << np.uint64(62)
    return key


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    sl = synth.make_libraries(seed=20260101, scale="full")
    ctx = _ffi.Context(0)
    casc = Cascade(ctx, sl.libs)
    reads = synth.make_reads_chunked(sl, n, seed=1000)
    raw = _ffi.DeviceReads.pack(ctx, reads)
    uniq = raw.collapse()
    useq = uniq.unpack()
    uniq.close(); raw.close()
    rng = np.random.default_rng(1)
    orders = {
        "collapse order": np.arange(len(useq)),
        "random order": rng.permutation(len(useq)),
        "sorted by the 13-base key": np.argsort(kmer_key(useq, 13), kind="stable"),
        "sorted by the 15-base key": np.argsort(kmer_key(useq, 15), kind="stable"),
    }
    # ... and sorted inside 2048-read tiles only (what a workgroup could do with its own survivors: its segment is ~2500 reads)
    k13 = kmer_key(useq, 13)
    tile = np.arange(len(useq)) // 2048
    orders["13-base key inside 2048-read tiles"] = np.lexsort((k13, tile))
    ref_ann = None
    for name, o in orders.items():
        dr = _ffi.DeviceReads.pack(ctx, useq.take(o))
        res = casc.run(dr)
        ann = res.fetch()
        res.close()
        inv = np.empty_like(o)
        inv[o] = np.arange(len(o))
        if ref_ann is None:
            ref_ann = ann
        else:
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
        print(f"{name:36s} wall {wall:7.3f} ms  " + "  ".join(f"{nm}={ms:.3f}" for nm, (ms, k) in sorted(recs.items()) if nm.endswith(".w1")), flush=True)
        dr.close()


if __name__ == "__main__":
    main()
