import sys, time
sys.path.insert(0, "/root/repo")
import mirge3_amd
from mirge3_amd import _ffi, synth
from mirge3_amd.cascade import Cascade, EXACT_PASS, ISO_PASS
sl = synth.make_libraries(seed=20260101, scale="full")
ctx = _ffi.Context(0)
casc = Cascade(ctx, sl.libs)
reads = synth.make_reads_chunked(sl, 10_000_000, seed=1000)
raw = _ffi.DeviceReads.pack(ctx, reads)
nm = len(sl.libs["mirna"])
def step():
    u = raw.collapse(); r = casc.run(u); _ffi.count_join(ctx, u, r, EXACT_PASS, ISO_PASS, nm); r.close(); u.close()
for _ in range(3): step()
for prof in (0, 1, 0, 1):
    ctx.profile(bool(prof)); ctx.profile_reset(); ctx.sync()
    t = time.perf_counter()
    for _ in range(20): step()
    ctx.sync()
    print("profiling", prof, "ms/step", round((time.perf_counter() - t) / 20 * 1e3, 3))
