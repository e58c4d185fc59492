import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["MIRGE_HOST_TIMING"] = "1"
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
def step():
    uniq, res = casc.collapse_and_run(raw)
    cls, ex, iso = _ffi.count_join(ctx, uniq, res, EXACT_PASS, ISO_PASS, n_mirna)
    res.close(); uniq.close()
for _ in range(20): step()
sys.stderr.write("=====MARK\n"); sys.stderr.flush()
t=time.perf_counter()
step()
sys.stderr.write("=====END %.1f us\n" % ((time.perf_counter()-t)*1e6))
