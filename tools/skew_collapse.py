"""Time the general collapse path (32-64 nt reads) on a uniform and on a Zipf-skewed group: python tools/skew_collapse.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import mirge3_amd
from mirge3_amd import _ffi
ctx = _ffi.Context(0)
rng = np.random.default_rng(5)
n = 400_000
tmpl = rng.integers(0, 4, size=(30_000, 40), dtype=np.uint8)
for name, idx in (("uniform", rng.integers(0, 30_000, size=n)),
                  ("zipf", np.minimum(rng.zipf(1.3, size=n) - 1, 29_999)),
                  ("one-third", np.where(rng.random(n) < 0.33, 0, rng.integers(0, 30_000, size=n)))):
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[tmpl[idx]]
    text = b"".join(b">r\n" + row.tobytes() + b"\n" for row in seqs[:n])
    raw, _ = _ffi.DeviceReads.parse(ctx, text, 2, 16)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); u = raw.collapse(); nu = len(u); cnt, first = u.counts(); dt = time.perf_counter() - t0
        u.close(); best = min(best, dt)
    print(f"{name:10s} reads {len(raw)} unique {nu} collapse {best*1e3:.3f} ms")
    raw.close()

# bursts of one <= 31-nt sequence behind distinct reads, in every writer's chunk: the level-1 regions of the partitioned
# key path overflow and the call is redone with chunk-sized regions
n = 10_000_000
tm = rng.integers(0, 4, size=(n // 2, 22), dtype=np.uint8)
idx = rng.integers(0, n // 2, size=n)
chunk = n // 256
idx[(np.arange(n) % chunk) >= chunk // 2] = 11
from mirge3_amd.seqio import FlatSeqs
seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[tm[idx]]
reads = FlatSeqs(seqs.reshape(-1).copy(), np.arange(n + 1, dtype=np.int64) * 22)
raw = _ffi.DeviceReads.pack(ctx, reads)
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); u = raw.collapse(); nu = len(u); dt = time.perf_counter() - t0
    u.close(); best = min(best, dt)
print(f"burst      reads {n} unique {nu} collapse {best*1e3:.3f} ms")
