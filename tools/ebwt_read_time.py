#!/usr/bin/env python3
"""How long `ebwt.read_ebwt` / `load_library_dir` take on a human-sized, ebwt-only library directory (what miRge3.0 ships: no
FASTA beside the indexes) -- CPU only, no GPU.  The indexes are written by tests/ebwt_writer.py (about 25 s for the 130 Mb mRNA
library: test infrastructure, pure Python), then read back and compared with the sequences they were written from.

  python tools/ebwt_read_time.py [--scale full]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", default="full")
    args = ap.parse_args()
    import mirge3_amd  # noqa: F401
    from mirge3_amd import ebwt, synth
    from mirge3_amd.seqio import index_basename, load_library_dir
    import ebwt_writer
    os.environ["MIRGE_LIB_CACHE"] = "0"
    sl = synth.make_libraries(seed=20260101, scale=args.scale)
    tmp = tempfile.mkdtemp(prefix="mirge_ebwt_time_", dir="/tmp")
    try:
        idx = os.path.join(tmp, "bench", "index.Libs")
        os.makedirs(idx)
        t = time.perf_counter()
        for key, lib in sl.libs.items():
            ebwt_writer.write_ebwt(os.path.join(idx, index_basename("bench", key, "miRBase")), lib.headers, lib.seqs.to_list())
        print(f"indexes written in {time.perf_counter() - t:.1f} s (tests/ebwt_writer.py)")
        for key, lib in sl.libs.items():
            t = time.perf_counter()
            got = ebwt.read_ebwt(os.path.join(idx, index_basename("bench", key, "miRBase")))
            dt = time.perf_counter() - t
            same = got.names == lib.names and got.seqs.data.shape == lib.seqs.data.shape and bool((got.seqs.data == lib.seqs.data).all())
            print(f"{key:14s} {len(lib):6d} references {lib.total_len / 1e6:8.2f} Mb  read_ebwt {dt:7.3f} s  {'same' if same else 'DIFFERS'}")
        t = time.perf_counter()
        load_library_dir(tmp, "bench", "miRBase")
        print(f"load_library_dir of the directory: {time.perf_counter() - t:.3f} s")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
