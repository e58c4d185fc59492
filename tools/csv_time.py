"""Time mirge_annotation_csv (host-only code) on synthetic arrays of the default sample's size: python tools/csv_time.py [outdir]
(MIRGE_HOST_TIMING=1 splits it into formatting and writing)"""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
import mirge3_amd
from mirge3_amd import _ffi
from mirge3_amd.seqio import FlatSeqs
rng = np.random.default_rng(1)
n = 4_200_000
lens = rng.integers(16, 31, size=n)
off = np.zeros(n + 1, np.int64); np.cumsum(lens, out=off[1:])
seqs = FlatSeqs(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(off[-1]))], off)
ps = rng.integers(-1, 9, size=n).astype(np.int8)
names = [FlatSeqs.from_list([f"hsa-miR-{i}-5p" for i in range(3000)]) for _ in range(9)]
ref = rng.integers(0, 3000, size=n).astype(np.int32)
counts = rng.integers(1, 50, size=(n, 1)).astype(np.uint32)
rows = rng.permutation(n).astype(np.int64)
hdr = "Sequence,annotFlag," + ",".join(f"c{i}" for i in range(9)) + ",s\n"
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
for rep in range(3):
    t = time.perf_counter()
    _ffi.annotation_csv(f"{out}/m.csv", f"{out}/u.csv", hdr, seqs, ps, ref, counts, rows, list(range(9)), 9, names)
    print(f"{time.perf_counter() - t:.3f} s", os.path.getsize(f"{out}/m.csv") + os.path.getsize(f"{out}/u.csv"))
