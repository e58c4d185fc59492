"""k_trim's launch time per 10 M 50-cycle records: the exact instance (one 3' adapter), the general instance at its 32-row height
(--match-read-wildcards set, the same reads kept) and, with MIRGE_TRIM_TALL=1 in the environment, at its 64-row height.
usage: python tools/trim_time.py [--reads N]"""
import argparse
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402
from mirge3_amd import _ffi, synth  # noqa: E402
from mirge3_amd.collapse import ILLUMINA_3P  # noqa: E402
from mirge3_amd.seqio import FlatSeqs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
a = ap.parse_args()
sl = synth.make_libraries(seed=7, scale="ci")
reads = synth.make_reads_chunked(sl, a.reads, seed=1000)
ad = np.frombuffer(ILLUMINA_3P.encode(), dtype=np.uint8)
L = reads.lengths
L2 = np.minimum(L + ad.shape[0], 50)
off2 = np.zeros(len(reads) + 1, dtype=np.int64)
np.cumsum(L2, out=off2[1:])
data2 = np.empty(int(off2[-1]), dtype=np.uint8)
rows = np.repeat(np.arange(len(reads), dtype=np.int64), L2)
within = np.arange(int(off2[-1]), dtype=np.int64) - off2[:-1][rows]
ins = within < L[rows]
data2[ins] = reads.data[(reads.offsets[:-1][rows] + within)[ins]]
data2[~ins] = ad[(within - L[rows])[~ins]]
text = bench.fastq_text(FlatSeqs(data2, off2))
ctx = _ffi.Context(0)
out = {}
for label, kw in (("exact", {}), ("general", {"read_wildcards": True})):
    trim = _ffi.MirgeTrim.make(adapter=ILLUMINA_3P, quality_back=10, count_per_modifier=False, **kw)
    for _ in range(2):
        r, _n = _ffi.DeviceReads.parse(ctx, text, 1, 16, trim); r.close()
    ctx.profile(True); ctx.profile_only("k_trim"); ctx.profile_reset()
    for _ in range(5):
        r, _n = _ffi.DeviceReads.parse(ctx, text, 1, 16, trim)
        kept = len(r); r.close()
    recs = [x for x in ctx.profile_records() if x[1]]
    ctx.profile(False); ctx.profile_only("")
    out[label] = {"kept": kept, "k_trim_ms": {n: round(ms / l, 3) for n, l, ms, u in recs}}
    print(label, out[label], flush=True)
e = sum(out["exact"]["k_trim_ms"].values()); g = sum(out["general"]["k_trim_ms"].values())
print(f"general / exact = {g / e:.2f}  (MIRGE_TRIM_TALL={os.environ.get('MIRGE_TRIM_TALL', '0')})")
