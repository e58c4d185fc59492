"""Throughput of TWO samples in flight (two contexts = two HIP streams, one host thread each) against one:
python tools/two_in_flight.py [reads]"""
import sys, time, threading
sys.path.insert(0, ".")
import mirge3_amd  # noqa: F401
from mirge3_amd import _ffi, synth
from mirge3_amd.cascade import Cascade, EXACT_PASS, ISO_PASS

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sl = synth.make_libraries(seed=20260101, scale="full")
n_mirna = len(sl.libs["mirna"])
lanes = []
for k in range(2):
    ctx = _ffi.Context(0)
    casc = Cascade(ctx, sl.libs, n_pass=9)
    raw = _ffi.DeviceReads.pack(ctx, synth.make_reads_chunked(sl, n_reads, seed=1000 + k))
    lanes.append((ctx, casc, raw))

def step(lane):
    ctx, casc, raw = lane
    uniq, res = casc.collapse_and_run(raw)
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, EXACT_PASS, ISO_PASS, n_mirna)
    res.close(); uniq.close()
    return cls

for lane in lanes:
    for _ in range(3): step(lane)
K = 40
t0 = time.perf_counter()
for _ in range(K): step(lanes[0])
one = (time.perf_counter() - t0) / K
def worker(lane, k):
    for _ in range(k): step(lane)
ths = [threading.Thread(target=worker, args=(lane, K)) for lane in lanes]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
two = (time.perf_counter() - t0) / (2 * K)
print(f"one sample at a time: {one*1e3:.3f} ms per sample = {n_reads/one/1e6:.0f} M reads/s")
print(f"two samples in flight: {two*1e3:.3f} ms per sample = {n_reads/two/1e6:.0f} M reads/s  ({one/two:.2f}x)")
