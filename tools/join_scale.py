"""k_join / k_tally with several sample columns (review: "global atomics per read; fine at S = 1, unmeasured at S = 8+").

S samples of `reads` raw reads each, collapsed together (U x S count matrix), annotated, then mirge_count_join and
mirge_variant_tally with every launch bracketed: kernel times for S = 1, 4, 8, 16.

  python tools/join_scale.py [reads_per_sample]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mirge3_amd  # noqa: E402,F401
from mirge3_amd import _ffi, a2i, synth  # noqa: E402
from mirge3_amd.cascade import Cascade, EXACT_PASS, ISO_PASS  # noqa: E402
from mirge3_amd.seqio import FlatSeqs  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    sl = synth.make_libraries(seed=20260101, scale="small")
    ctx = _ffi.Context(0)
    casc = Cascade(ctx, sl.libs)
    n_mirna = len(sl.libs["mirna"])
    for S in (1, 4, 8, 16):
        samples = [synth.make_reads_chunked(sl, n, seed=500 + s) for s in range(S)]
        data = np.concatenate([x.data for x in samples])
        lens = np.concatenate([x.lengths for x in samples])
        off = np.zeros(lens.shape[0] + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        raw = _ffi.DeviceReads.pack(ctx, FlatSeqs(data, off))
        sid = np.repeat(np.arange(S, dtype=np.int32), [len(x) for x in samples]) if S > 1 else None
        uniq = raw.collapse(sid, S)
        res = casc.run(uniq)
        _ffi.count_join(ctx, uniq, res, EXACT_PASS, ISO_PASS, n_mirna)  # warm
        a2i.tally(casc, uniq, res)
        ctx.profile(True); ctx.profile_only(""); ctx.profile_reset()
        for _ in range(5):
            cls, ex, iso = _ffi.count_join(ctx, uniq, res, EXACT_PASS, ISO_PASS, n_mirna)
            a2i.tally(casc, uniq, res)
        ctx.sync()
        recs = {nm: ms / l for nm, l, ms, u in ctx.profile_records() if l and nm.startswith(("k_join", "k_tally", "k_member"))}
        ctx.profile(False)
        assert int(cls.sum()) + 0 <= S * n
        print(f"S = {S:2d}  U = {len(uniq):9d}  " + "  ".join(f"{k} {v:.4f} ms" for k, v in sorted(recs.items())), flush=True)
        res.close(); uniq.close(); raw.close()


if __name__ == "__main__":
    main()
