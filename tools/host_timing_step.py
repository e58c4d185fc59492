"""Where the host spends a step (bench.py's step = collapse_and_run + count_join + two closes), microseconds.
  python tools/host_timing_step.py        (MIRGE_HOST_TIMING=1 adds the library's own laps on stderr)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mirge3_amd
from mirge3_amd import _ffi, synth
from mirge3_amd.cascade import Cascade, EXACT_PASS, ISO_PASS
sl = synth.make_libraries(seed=20260101, scale="full")
ctx = _ffi.Context(0)
casc = Cascade(ctx, sl.libs)
reads = synth.make_reads_chunked(sl, 10_000_000, seed=1000)
raw = _ffi.DeviceReads.pack(ctx, reads)
n_mirna = len(sl.libs["mirna"])
import gc; gc.collect(); gc.freeze()
acc = np.zeros(4)
def step(rec):
    t0 = time.perf_counter()
    uniq, res = casc.collapse_and_run(raw)
    t1 = time.perf_counter()
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, EXACT_PASS, ISO_PASS, n_mirna)
    t2 = time.perf_counter()
    res.close(); uniq.close()
    t3 = time.perf_counter()
    if rec: acc[:3] += (t1 - t0, t2 - t1, t3 - t2)
for _ in range(50): step(False)
N = 300
t = time.perf_counter()
for _ in range(N): step(True)
tot = (time.perf_counter() - t) / N * 1e6
a = acc / N * 1e6
print(f"step {tot:.1f} us: collapse_and_run call {a[0]:.1f}, count_join call {a[1]:.1f}, closes {a[2]:.1f}, python between {tot - a[:3].sum():.1f}")
