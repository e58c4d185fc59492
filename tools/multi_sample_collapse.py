#!/usr/bin/env python3
"""What the joint collapse of S samples in ONE process costs against S per-sample collapses (the CLI's route for several files is the first:
fastpath.run concatenates the parsed samples and calls mirge_collapse with sample ids, which takes the general -- global-atomic -- path).
  python tools/multi_sample_collapse.py [--samples 4] [--reads 10000000] [--pool 1250000]"""
import argparse
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--pool", type=int, default=1_250_000)
    args = ap.parse_args()
    import mirge3_amd  # noqa: F401
    from mirge3_amd import _ffi, synth
    from mirge3_amd.cascade import Cascade, EXACT_PASS, ISO_PASS
    sl = synth.make_libraries(seed=20260101, scale="full")
    ctx = _ffi.Context(0)
    casc = Cascade(ctx, sl.libs, n_pass=9)
    n_mirna = len(sl.libs["mirna"])
    for label, kw in (("zipf", dict(pool=args.pool)), ("default", {})):
        raws = []
        for s in range(args.samples):
            reads = synth.make_reads(sl, args.reads, seed=2000 + s, **kw) if kw else synth.make_reads_chunked(sl, args.reads, seed=1000 + s)
            raws.append(_ffi.DeviceReads.pack(ctx, reads))
        S = len(raws)
        sid = np.repeat(np.arange(S, dtype=np.int32), [len(r) for r in raws])

        def per_sample():
            for r in raws:
                u, res = casc.collapse_and_run(r)
                _ffi.count_join(ctx, u, res, EXACT_PASS, ISO_PASS, n_mirna)
                res.close(); u.close()

        def joint():
            allr = _ffi.DeviceReads.concat(ctx, raws)
            t1 = time.perf_counter()
            u = allr.collapse(sid, S)
            ctx.sync()
            t2 = time.perf_counter()
            res = casc.run(u)
            _ffi.count_join(ctx, u, res, EXACT_PASS, ISO_PASS, n_mirna)
            nu = len(u)
            res.close(); u.close(); allr.close()
            return t2 - t1, nu
        def merged():
            t1 = time.perf_counter()
            dicts = [r.collapse() for r in raws]
            u = _ffi.DeviceReads.merge(ctx, dicts)
            ctx.sync()
            t2 = time.perf_counter()
            for d in dicts:
                d.close()
            res = casc.run(u)
            _ffi.count_join(ctx, u, res, EXACT_PASS, ISO_PASS, n_mirna)
            nu = len(u)
            res.close(); u.close()
            return t2 - t1, nu
        for f in (per_sample, joint, merged):
            f(); f()
        t = time.perf_counter(); k = 0
        while k < 5 or time.perf_counter() - t < 0.5:
            per_sample(); k += 1
        a = (time.perf_counter() - t) / k
        t = time.perf_counter(); kj = 0; tc = 0.0
        while kj < 5 or time.perf_counter() - t < 0.5:
            cj, nu = joint(); tc += cj; kj += 1
        b = (time.perf_counter() - t) / kj
        t = time.perf_counter(); k = 0; tm = 0.0
        while k < 5 or time.perf_counter() - t < 0.5:
            cm, num = merged(); tm += cm; k += 1
        m = (time.perf_counter() - t) / k
        assert num == nu, (num, nu)
        print(f"{label:8s} merged dictionaries (mirge_collapse_merge): run {m * 1e3:.3f} ms ({m * 1e3 / S:.3f} per sample; per-sample collapses + merge {tm / k * 1e3:.3f} ms)", flush=True)
        print(f"{label:8s} {S} samples x {args.reads / 1e6:g} M reads: per-sample steps {a * 1e3:.3f} ms ({a * 1e3 / S:.3f} per sample); joint run {b * 1e3:.3f} ms "
              f"({b * 1e3 / S:.3f} per sample; its collapse {tc / kj * 1e3:.3f} ms, {nu / 1e6:.2f} M unique reads of the union)", flush=True)
        for r in raws:
            r.close()


if __name__ == "__main__":
    main()
