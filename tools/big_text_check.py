"""A FASTQ text larger than one mirge_reads_parse call takes (8 GiB), on a GPU box: 17 copies of a 10 M-read sample = 8.75 GB,
170 M records, parsed in parts of whole records (collapse.TextRecordStream) and collapsed; every count must be 17 x the one
sample's.  Prints one JSON line.  Refuses to start with less than 96 GB of host memory available."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mirge3_amd  # noqa: E402,F401
from mirge3_amd import _ffi, collapse, synth  # noqa: E402
from bench import fastq_text  # noqa: E402


def available_gb():
    avail = None
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable"):
            avail = int(ln.split()[1]) / 1e6
    for f in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(f).read().strip()
            if v != "max":
                avail = min(avail, int(v) / 1e9)
        except OSError:
            pass
    return avail


def main():
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    gb = available_gb()
    if gb is None or gb < 96:
        print(json.dumps({"skipped": f"{gb} GB of host memory available"}))
        return
    ctx = _ffi.Context(0)
    sl = synth.make_libraries(seed=20260101, scale="ci")
    reads = synth.make_reads_chunked(sl, 10_000_000, seed=11)
    one = fastq_text(reads)
    del reads
    raw1, n1 = collapse.parse_sample(ctx, one, 16, None, None)
    u1 = raw1.collapse()
    c1, f1 = u1.counts()
    big = np.tile(one, copies)
    out = {"copies": copies, "text_GB": round(big.size / 1e9, 2), "host_available_GB": round(gb)}
    try:
        _ffi.DeviceReads.parse(ctx, big, 1, 16, None)
        out["whole_parse"] = "accepted"
    except RuntimeError as e:
        out["whole_parse"] = "refused: " + str(e)[-80:]
    t0 = time.perf_counter()
    tm = {}
    raw, n = collapse.parse_sample(ctx, collapse.TextRecordStream(big), 16, None, None, timings=tm)
    out["parse_in_parts_s"] = round(time.perf_counter() - t0, 2)
    out["parts"] = tm["gz_pieces"]
    out["records"] = n
    t0 = time.perf_counter()
    u = raw.collapse()
    ctx.sync()
    out["collapse_s"] = round(time.perf_counter() - t0, 3)
    c, f = u.counts()
    o1, o = np.argsort(f1, kind="stable"), np.argsort(f, kind="stable")
    out["ok"] = bool(n == copies * n1 and len(raw) == copies * len(raw1) and len(u) == len(u1)
                     and np.array_equal(c[o].astype(np.int64), copies * c1[o1].astype(np.int64)) and np.array_equal(f[o], f1[o1]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
