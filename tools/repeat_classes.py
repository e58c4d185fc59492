#!/usr/bin/env python3
"""Which repeat-derived read class costs the bulk cascade what, on the repeat-rich libraries (synth.make_libraries(repeats=True)):
the default mix plus ONE of poly / simple / alu at a time, then all three (synth.REPEAT_MIX).  Prints the step, the bulk kernel's
time and how its workgroups' times spread.   python tools/repeat_classes.py [--reads 10000000] [--scale full]"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--scale", default="full")
    args = ap.parse_args()
    import mirge3_amd  # noqa: F401
    from mirge3_amd import _ffi, synth
    from mirge3_amd.cascade import Cascade, EXACT_PASS, ISO_PASS
    sl = synth.make_libraries(seed=20260101, scale=args.scale, repeats=True)
    ctx = _ffi.Context(0)
    casc = Cascade(ctx, sl.libs, n_pass=9)
    n_mirna = len(sl.libs["mirna"])
    base = dict(synth.DEFAULT_MIX)
    mixes = {"none": base, "poly": dict(base, poly=0.06), "simple": dict(base, simple=0.03), "alu": dict(base, alu=0.06),
             "all": synth.REPEAT_MIX}
    for name, mix in mixes.items():
        reads = synth.make_reads_chunked(sl, args.reads, seed=4000, mix=mix)
        raw = _ffi.DeviceReads.pack(ctx, reads)

        def one():
            uq, rs = casc.collapse_and_run(raw)
            _ffi.count_join(ctx, uq, rs, EXACT_PASS, ISO_PASS, n_mirna)
            n = len(uq)
            ps = None
            rs.close(); uq.close()
            return n
        for _ in range(3):
            u = one()
        ctx.profile(True); ctx.profile_only(""); ctx.profile_reset()
        for _ in range(3):
            one()
        recs = [r for r in ctx.profile_records() if r[1]]
        wg = _ffi.cascade_wg_times(ctx)
        ctx.profile(False)
        top = sorted(((nm, ms / l) for nm, l, ms, _ in recs), key=lambda x: -x[1])[:4]
        t = time.perf_counter()
        k = 0
        while k < 5 or time.perf_counter() - t < 0.3:
            one(); k += 1
        dt = (time.perf_counter() - t) / k
        print(f"{name:7s} U {u / 1e6:.2f} M  step {dt * 1e3:8.3f} ms  " + "  ".join(f"{nm} {v:.3f}" for nm, v in top)
              + (f"  | bulk workgroups: median {np.median(wg):.3f} p90 {np.percentile(wg, 90):.3f} p99 {np.percentile(wg, 99):.3f} max {wg.max():.3f} mean {wg.mean():.3f}" if wg is not None else ""),
              flush=True)
        raw.close()


if __name__ == "__main__":
    main()
