"""ctypes binding of ``csrc/libmirge_native.so`` (C ABI: ``include/mirge_native.h``).

This is the whole boundary between the Python host code and the HIP kernels: plain pointers and
sizes, numpy buffers in and out.  There is no fallback: if the shared library is missing, or no
MI355X is visible, the first call raises.
"""
from __future__ import annotations

import ctypes as C
import weakref
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .seqio import FlatSeqs

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIRGE_NATIVE_SO: A/B builds of the same library when tuning kernels (never a different backend)
SO_PATH = os.environ.get("MIRGE_NATIVE_SO") or os.path.join(_HERE, "csrc", "libmirge_native.so")

EXPORTS = [
    "mirge_last_error", "mirge_device_count", "mirge_gz_inflate", "mirge_gz_inflate_progress", "mirge_ctx_create", "mirge_ctx_destroy", "mirge_ctx_sync",
    "mirge_lib_create", "mirge_lib_create_packed", "mirge_lib_packed_sizes", "mirge_lib_packed_copy", "mirge_lib_destroy", "mirge_lib_n_refs", "mirge_lib_device_bytes", "mirge_lib_prepare",
    "mirge_reads_pack", "mirge_reads_parse", "mirge_reads_parse_trim", "mirge_reads_parse_umi", "mirge_reads_concat", "mirge_reads_destroy", "mirge_reads_count", "mirge_reads_total_bases",
    "mirge_reads_n_samples", "mirge_reads_group_counts", "mirge_reads_iupac_seen", "mirge_reads_unpack", "mirge_collapse", "mirge_collapse_weighted", "mirge_collapse_merge", "mirge_collapse_fetch", "mirge_collapse_order", "mirge_collapse_order_sorted", "mirge_collapse_nonzero",
    "mirge_reads_set_counts", "mirge_cascade_run", "mirge_collapse_cascade", "mirge_result_fetch", "mirge_result_destroy",
    "mirge_count_join", "mirge_count_join_host", "mirge_annotation_csv", "mirge_annotation_csv_device", "mirge_variant_tally", "mirge_isomir_type", "mirge_gff_write", "mirge_gff_write_device", "mirge_ctx_timer_start", "mirge_ctx_timer_stop", "mirge_ctx_profile_enable",
    "mirge_reads_range_sample", "mirge_reads_range_split", "mirge_annotation_csv_device_sizes", "mirge_annotation_csv_device_at",
    "mirge_cascade_prepare", "mirge_cascade_walks", "mirge_cascade_wg_times", "mirge_ctx_profile_only", "mirge_ctx_profile_units", "mirge_ctx_profile_reset", "mirge_ctx_profile_count", "mirge_ctx_profile_get",
]


class MirgePolicy(C.Structure):
    """``mirge_policy`` of include/mirge_native.h."""
    _fields_ = [(n, C.c_int32) for n in (
        "mode", "mm", "seedlen", "maxtotal", "trim5", "trim3", "ttail", "len_lt", "len_gt", "reserved")]


class MirgeTrim(C.Structure):
    """``mirge_trim`` of include/mirge_native.h: the cutadapt modifier chain of digest.py:59-101."""
    _fields_ = [("nextseq_cutoff", C.c_int32), ("quality_front", C.c_int32), ("quality_back", C.c_int32),
                ("phred_base", C.c_int32), ("adapter", C.c_char_p), ("adapter_len", C.c_int32), ("min_overlap", C.c_int32),
                ("error_rate", C.c_double), ("trim_n", C.c_int32), ("n_cut", C.c_int32), ("cut", C.c_int32 * 2),
                ("count_per_modifier", C.c_int32), ("adapter_front", C.c_int32), ("adapter2", C.c_char_p),
                ("adapter2_len", C.c_int32), ("adapter2_front", C.c_int32), ("times", C.c_int32), ("no_indels", C.c_int32),
                ("match_read_wildcards", C.c_int32), ("no_adapter_wildcards", C.c_int32), ("action_none", C.c_int32),
                ("adapter_anchored", C.c_int32), ("adapter2_anchored", C.c_int32), ("linked", C.c_int32), ("linked_required", C.c_int32)]

    @staticmethod
    def make(adapter: Optional[str] = None, quality_back: int = -1, quality_front: int = 0, nextseq: int = -1,
             phred_base: int = 33, min_overlap: int = 3, error_rate: float = 0.12, trim_n: bool = False,
             cut: Sequence[int] = (), count_per_modifier: bool = True, front: bool = False,
             adapter2: Optional[str] = None, front2: bool = False, times: int = 1, indels: bool = True,
             read_wildcards: bool = False, adapter_wildcards: bool = True, action: str = "trim",
             anchored: bool = False, anchored2: bool = False, linked: bool = False, front_required: bool = True,
             back_required: bool = False) -> "MirgeTrim":
        t = MirgeTrim()
        t.nextseq_cutoff, t.quality_front, t.quality_back, t.phred_base = nextseq, quality_front, quality_back, phred_base
        a = adapter.encode() if adapter else None
        t.adapter, t.adapter_len = a, (len(a) if a else 0)
        t.min_overlap, t.error_rate, t.trim_n = min_overlap, error_rate, 1 if trim_n else 0
        cut = [int(x) for x in cut]
        t.n_cut = len(cut)
        for k, v in enumerate(cut[:2]):
            t.cut[k] = v
        t.count_per_modifier = 1 if count_per_modifier else 0
        t.adapter_front = 1 if front else 0
        a2 = adapter2.encode() if adapter2 else None
        t.adapter2, t.adapter2_len, t.adapter2_front = a2, (len(a2) if a2 else 0), 1 if front2 else 0
        t.times, t.no_indels = int(times), 0 if indels else 1
        t.match_read_wildcards, t.no_adapter_wildcards = 1 if read_wildcards else 0, 0 if adapter_wildcards else 1
        if action not in ("trim", "none"):
            raise NotImplementedError("--action mask / lowercase change the letters of a read, not its bounds: not part of the MI355X path")
        t.action_none = 1 if action == "none" else 0
        t.adapter_anchored, t.adapter2_anchored = 1 if anchored else 0, 1 if anchored2 else 0
        t.linked = 1 if linked else 0
        t.linked_required = (1 if front_required else 0) | (2 if back_required else 0)
        return t


class MirgeUmi(C.Structure):
    """``mirge_umi`` of include/mirge_native.h: ``-umi f,b`` [``--qiagenumi``] [``-udd``] (digest.py:164-205,305-315,334-365)."""
    _fields_ = [("front", C.c_int32), ("back", C.c_int32), ("qiagen", C.c_int32), ("dedup", C.c_int32)]

    @staticmethod
    def make(front: int, back: int, qiagen: bool = False, dedup: bool = False) -> "MirgeUmi":
        u = MirgeUmi()
        u.front, u.back, u.qiagen, u.dedup = int(front), int(back), 1 if qiagen else 0, 1 if dedup else 0
        return u


_lib = None
_live = []  # weak references to every wrapper, closed in dependency order at interpreter exit


def load() -> C.CDLL:
    """Load the shared library (once).  Raises if it was not built -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  mirge3.0_amd has no CPU fallback.")
    lib = C.CDLL(SO_PATH)
    lib.mirge_last_error.restype = C.c_char_p
    for name in ("mirge_lib_n_refs", "mirge_lib_device_bytes", "mirge_reads_count", "mirge_reads_total_bases"):
        getattr(lib, name).restype = C.c_int64
    for name in ("mirge_lib_destroy", "mirge_reads_destroy", "mirge_result_destroy", "mirge_ctx_destroy"):
        getattr(lib, name).restype = None
    _lib = lib
    return lib


def _track(obj):
    _live.append(weakref.ref(obj))
    if len(_live) > 4096:  # long-running callers create two handles per sample: drop the dead references
        _live[:] = [r for r in _live if r() is not None]


def _shutdown():
    """Close device handles before the interpreter and the HIP runtime tear down: result/read sets,
    then libraries, then contexts (a handle freed after its context is gone would crash at exit)."""
    objs = [r() for r in _live]
    objs = [o for o in objs if o is not None]
    for kind in (CascadeResult, DeviceReads, DeviceLibrary, Context):
        for o in objs:
            if isinstance(o, kind):
                try:
                    o.close()
                except Exception:
                    pass


import atexit  # noqa: E402
atexit.register(_shutdown)


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().mirge_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def _p(a: Optional[np.ndarray]):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def gz_inflate(data, threads: int = 0) -> Optional[np.ndarray]:
    """The text of a whole .gz file (``data``: its bytes) inflated on all host cores (``mirge_gz_inflate``), or None when the
    file is not of a kind that route takes -- the caller then inflates it serially.  Verified against the file's CRC-32."""
    buf = np.frombuffer(data, dtype=np.uint8)
    if buf.size < 18:
        return None
    isize = int.from_bytes(bytes(buf[-4:]), "little")  # the LAST member's length mod 2^32: exact for a one-member file below 4 GiB
    # room: that length, or -- the file may hold several members (merged lanes, BGZF), whose lengths only the members know --
    # twelve times the compressed size (FASTQ deflates 4-6 x; untouched pages of the buffer cost nothing).  Too little room is
    # reported (-2) and the caller streams the file instead.
    cap = max(isize, 12 * buf.size) + 65536
    if cap > (1 << 37):
        return None
    try:
        out = np.empty(cap, dtype=np.uint8)
    except MemoryError:  # (a host that does not overcommit): the streamed route needs pieces only
        return None
    n = C.c_int64(0)
    rc = load().mirge_gz_inflate(_p(buf), C.c_int64(buf.size), _p(out), C.c_int64(cap), C.byref(n), C.c_int32(threads or gz_threads()))
    if rc != 0:
        return None
    return out[: n.value]


# The text buffer of the last finished GzInflation, kept for the next one: a fresh buffer costs a sample its page faults while it
# inflates (the inflater's floor, profiles/README.md round 4) and an munmap of half a gigabyte when it is dropped -- 0.03-0.05 s of
# a 0.15 s sample.  One buffer, at most MIRGE_GZ_KEEP_BYTES (default 16 GiB of address space; what a sample touched stays resident).
_gz_kept: list = []          # (popped and appended by read-ahead threads: list.pop / append are atomic, the emptiness test is not)
_GZ_KEEP_BYTES = int(os.environ.get("MIRGE_GZ_KEEP_BYTES", str(16 << 30)))
import threading as _threading
_gz_keep_lock = _threading.Lock()


def gz_threads() -> int:
    """host threads one inflation takes: MIRGE_GZ_THREADS, else the cores divided among the ranks of this node (one process per GPU:
    eight samples inflate side by side), at most 64 (beyond, nothing is gained)"""
    env = os.environ.get("MIRGE_GZ_THREADS")
    if env:
        return max(1, int(env))
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    return max(1, min(64, (os.cpu_count() or 1) // ranks))


class GzInflation:
    """``mirge_gz_inflate_progress`` running on a thread of its own: ``out[:done()]`` is text that will not change any more, whole
    before ``wait()`` returns -- True when the file was inflated and its CRC-32 matched, False when this route does not take the
    file (then whatever was read ahead must be dropped and the file inflated the serial way)."""

    def __init__(self, data, threads: int = 0):
        import threading
        self.buf = data if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
        self.out = None
        self.ok = None
        self._n = C.c_int64(0)
        self._progress = C.c_int64(0)
        self._thread = None
        if self.buf.size < 18:
            self.ok = False
            return
        isize = int.from_bytes(bytes(self.buf[-4:]), "little")
        cap = max(isize, 12 * int(self.buf.size)) + 65536  # (see gz_inflate)
        if cap > (1 << 37):
            self.ok = False
            return
        try:
            kept = _gz_kept.pop()
        except IndexError:  # none kept, or another read-ahead thread took it between a test and the pop
            kept = None
        if kept is not None and kept.size >= cap:
            self.out = kept
            cap = int(kept.size)
        else:
            del kept
            try:
                self.out = np.empty(cap, dtype=np.uint8)
            except MemoryError:
                self.ok = False
                return
        lib = load()

        def run():
            rc = lib.mirge_gz_inflate_progress(_p(self.buf), C.c_int64(self.buf.size), _p(self.out), C.c_int64(cap), C.byref(self._n),
                                               C.c_int32(threads or gz_threads()), C.byref(self._progress))
            self.ok = rc == 0

        # (not a daemon: the interpreter waits for it at exit -- a fraction of a second -- instead of freeing the buffers under it)
        self._thread = threading.Thread(target=run, name="mirge-gz-inflate", daemon=False)
        self._thread.start()

    def done(self) -> int:
        """bytes of ``out`` that are final so far"""
        return int(self._progress.value)

    def running(self) -> bool:
        return self._thread is not None and self._thread.is_alive()

    def wait(self) -> bool:
        if self._thread is not None:
            self._thread.join()
        return bool(self.ok)

    def text(self) -> np.ndarray:
        """the whole text (after ``wait()`` returned True)"""
        return self.out[: self._n.value]

    def release(self):
        """the caller is done with ``out`` and every view of it: the buffer is kept for the next inflation"""
        self.wait()
        out, self.out = self.out, None
        if out is not None and out.size <= _GZ_KEEP_BYTES:
            with _gz_keep_lock:
                if not _gz_kept:
                    _gz_kept.append(out)


class Context:
    """One GPU + one HIP stream (``mirge_ctx``)."""

    def __init__(self, device: int = 0, stream: int = 0):
        lib = load()
        if lib.mirge_device_count() <= 0:
            raise RuntimeError("no HIP device visible: mirge3.0_amd needs an MI355X (no CPU fallback)")
        self._h = C.c_void_p()
        _check(lib.mirge_ctx_create(C.c_int(device), C.c_void_p(stream), C.byref(self._h)), "mirge_ctx_create")
        self.device = device
        _track(self)

    def close(self):
        if self._h:
            # whatever still lives in this context goes first -- results, read sets, libraries: their device blocks belong to the
            # context's pool, and a handle destroyed after its context (a reference kept by a traceback, a cycle the collector
            # reaches late) would free into released memory
            objs = [o for o in (r() for r in _live) if o is not None and getattr(o, "ctx", None) is self]
            for kind in (CascadeResult, DeviceReads, DeviceLibrary):
                for o in objs:
                    if isinstance(o, kind):
                        o.close()
            load().mirge_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(load().mirge_ctx_sync(self._h), "mirge_ctx_sync")

    # ---- measurement
    def timer_start(self):
        _check(load().mirge_ctx_timer_start(self._h), "mirge_ctx_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_double()
        _check(load().mirge_ctx_timer_stop(self._h, C.byref(ms)), "mirge_ctx_timer_stop")
        return ms.value

    def profile(self, on: bool):
        _check(load().mirge_ctx_profile_enable(self._h, C.c_int32(1 if on else 0)), "profile_enable")

    def profile_only(self, substr: str = ""):
        """bracket only the launches whose name contains ``substr`` ('' = all)"""
        _check(load().mirge_ctx_profile_only(self._h, substr.encode()), "profile_only")

    def profile_units(self, on: bool):
        """False: bracketed cascade launches are timed, their per-pass read counts are not copied back any more"""
        _check(load().mirge_ctx_profile_units(self._h, C.c_int32(1 if on else 0)), "profile_units")

    def profile_reset(self):
        _check(load().mirge_ctx_profile_reset(self._h), "profile_reset")

    def profile_records(self):
        """[(kernel name, launches, total ms, units)]"""
        lib = load()
        out = []
        for i in range(lib.mirge_ctx_profile_count(self._h)):
            name = C.create_string_buffer(64)
            launches, ms, units = C.c_int64(), C.c_double(), C.c_double()
            _check(lib.mirge_ctx_profile_get(self._h, C.c_int32(i), name, C.c_int32(64), C.byref(launches),
                                             C.byref(ms), C.byref(units)), "profile_get")
            out.append((name.value.decode(), launches.value, ms.value, units.value))
        return out


class DeviceLibrary:
    """A reference library packed and indexed in HBM (``mirge_lib``)."""

    def __init__(self, ctx: Context, seqs: FlatSeqs):
        self.ctx = ctx
        self._h = C.c_void_p()
        packed = getattr(seqs, "packed", None)
        if packed is not None:  # a cached library (libcache.PackedSeqs): the image itself, no letters, no packing
            T = np.ascontiguousarray(packed["T"], dtype=np.uint64)
            inv = np.ascontiguousarray(packed["inv"], dtype=np.uint64)
            rs = np.ascontiguousarray(packed["ref_start"], dtype=np.uint32)
            _check(load().mirge_lib_create_packed(ctx._h, _p(T), C.c_int64(T.shape[0]), _p(inv), C.c_int64(inv.shape[0]), _p(rs),
                                                  C.c_int64(len(seqs)), C.c_uint64(int(packed["total"])), C.c_int32(int(packed["kmax"])),
                                                  C.c_uint64(int(packed["valid_positions"])), C.byref(self._h)), "mirge_lib_create_packed")
        else:
            data = np.ascontiguousarray(seqs.data, dtype=np.uint8)
            off = np.ascontiguousarray(seqs.offsets, dtype=np.int64)
            _check(load().mirge_lib_create(ctx._h, _p(data), _p(off), C.c_int64(len(seqs)), C.byref(self._h)),
                   "mirge_lib_create")
        _track(self)

    def packed_image(self) -> dict:
        """the library's packed image (2-bit text, invalid-base bitmap, reference starts) for the cache next to the index"""
        sizes = (C.c_int64 * 4)()
        kmax = C.c_int32()
        _check(load().mirge_lib_packed_sizes(self._h, sizes, C.byref(kmax)), "mirge_lib_packed_sizes")
        T = np.empty(int(sizes[0]), dtype=np.uint64)
        inv = np.empty(int(sizes[1]), dtype=np.uint64)
        rs = np.empty(self.n_refs + 1, dtype=np.uint32)
        _check(load().mirge_lib_packed_copy(self._h, _p(T), _p(inv), _p(rs)), "mirge_lib_packed_copy")
        return {"T": T, "inv": inv, "ref_start": rs, "total": int(sizes[2]), "kmax": int(kmax.value), "valid_positions": int(sizes[3])}

    @property
    def n_refs(self) -> int:
        return load().mirge_lib_n_refs(self._h)

    @property
    def device_bytes(self) -> int:
        return load().mirge_lib_device_bytes(self._h)

    def prepare(self, k: int):
        _check(load().mirge_lib_prepare(self._h, C.c_int32(k)), "mirge_lib_prepare")

    def close(self):
        if self._h:
            load().mirge_lib_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceReads:
    """A device-resident packed read set (``mirge_reads``), optionally with a count matrix."""

    def __init__(self, ctx: Context, handle: C.c_void_p):
        self.ctx = ctx
        self._h = handle
        _track(self)

    @staticmethod
    def pack(ctx: Context, reads: FlatSeqs) -> "DeviceReads":
        data = np.ascontiguousarray(reads.data, dtype=np.uint8)
        off = np.ascontiguousarray(reads.offsets, dtype=np.int64)
        h = C.c_void_p()
        _check(load().mirge_reads_pack(ctx._h, _p(data), _p(off), C.c_int64(len(reads)), C.byref(h)),
               "mirge_reads_pack")
        return DeviceReads(ctx, h)

    @staticmethod
    def parse(ctx: Context, text, fmt: int = 0, min_len: int = 0, trim: Optional["MirgeTrim"] = None):
        """FASTQ / single-line FASTA / one-sequence-per-line TEXT (bytes or a uint8 array) -> (DeviceReads of the
        records with len >= min_len, number of records seen).  Parsed -- and, with ``trim``, trimmed (quality, 3'
        adapter, N ends, cuts: ``mirge_reads_parse_trim``) -- on the GPU: no per-read host work."""
        buf = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else \
            np.ascontiguousarray(text, dtype=np.uint8)
        h = C.c_void_p()
        nrec = C.c_int64()
        _check(load().mirge_reads_parse_trim(ctx._h, _p(buf) if buf.size else C.c_void_p(0), C.c_int64(buf.size), C.c_int32(fmt),
                                             C.c_int32(min_len), C.byref(trim) if trim is not None else C.c_void_p(0),
                                             C.byref(h), C.byref(nrec)), "mirge_reads_parse")
        return DeviceReads(ctx, h), int(nrec.value)

    @staticmethod
    def parse_umi(ctx: Context, text, fmt: int, min_len: int, trim: Optional["MirgeTrim"], umi: "MirgeUmi"):
        """``parse`` with the reference's UMI handling (``mirge_reads_parse_umi``) -> (raw INSERTS, records seen, the
        distinct UMI-tagged reads with their counts -- a collapse result -- or None without ``-udd``).  With ``-udd``
        the inserts are one per distinct tagged read, in the order the tagged reads first appeared."""
        buf = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else \
            np.ascontiguousarray(text, dtype=np.uint8)
        h, ht = C.c_void_p(), C.c_void_p()
        nrec = C.c_int64()
        _check(load().mirge_reads_parse_umi(ctx._h, _p(buf) if buf.size else C.c_void_p(0), C.c_int64(buf.size), C.c_int32(fmt),
                                            C.c_int32(min_len), C.byref(trim) if trim is not None else C.c_void_p(0),
                                            C.byref(umi), C.byref(h), C.byref(nrec), C.byref(ht)), "mirge_reads_parse_umi")
        return DeviceReads(ctx, h), int(nrec.value), (DeviceReads(ctx, ht) if ht.value else None)

    @staticmethod
    def concat(ctx: Context, parts: Sequence["DeviceReads"]) -> "DeviceReads":
        """Raw read sets appended in the order given (one per sample, before the joint collapse)."""
        arr = (C.c_void_p * len(parts))(*[p._h for p in parts])
        h = C.c_void_p()
        _check(load().mirge_reads_concat(ctx._h, arr, C.c_int32(len(parts)), C.byref(h)), "mirge_reads_concat")
        return DeviceReads(ctx, h)

    def __len__(self) -> int:
        return load().mirge_reads_count(self._h)

    @property
    def n_samples(self) -> int:
        return load().mirge_reads_n_samples(self._h)

    def group_counts(self) -> np.ndarray:
        """reads per storage group: width classes (1, 2, 4, 8 words) without an ambiguous call, then the same with an N"""
        out = np.zeros(16, dtype=np.int64)
        n = load().mirge_reads_group_counts(self._h, _p(out), C.c_int32(16))
        return out[:n]

    @property
    def iupac_seen(self) -> bool:
        """some read held an IUPAC ambiguity code other than N (packed and printed as N)"""
        return load().mirge_reads_iupac_seen(self._h) == 1

    @staticmethod
    def merge(ctx: "Context", parts: Sequence["DeviceReads"]) -> "DeviceReads":
        """The sample matrix of several samples from their per-sample dictionaries (``mirge_collapse_merge``): unique reads of the
        union, one count column per part, all on the device."""
        arr = (C.c_void_p * len(parts))(*[p._h for p in parts])
        h, nu = C.c_void_p(), C.c_int64()
        _check(load().mirge_collapse_merge(ctx._h, arr, C.c_int32(len(parts)), C.byref(h), C.byref(nu)), "mirge_collapse_merge")
        return DeviceReads(ctx, h)

    def unpack(self) -> FlatSeqs:
        n = len(self)
        off = np.zeros(n + 1, dtype=np.int64)
        data = np.empty(max(load().mirge_reads_total_bases(self._h), 1), dtype=np.uint8)
        _check(load().mirge_reads_unpack(self.ctx._h, self._h, _p(data), _p(off)), "mirge_reads_unpack")
        return FlatSeqs(data[:off[-1]].copy(), off)

    def collapse(self, sample_ids: Optional[np.ndarray] = None, n_samples: int = 1,
                 weights: Optional[np.ndarray] = None) -> "DeviceReads":
        """``weights`` (uint32 per read): read i stands for that many copies (``mirge_collapse_weighted``)."""
        sid = None if sample_ids is None else np.ascontiguousarray(sample_ids, dtype=np.int32)
        h = C.c_void_p()
        nu = C.c_int64()
        if weights is None:
            _check(load().mirge_collapse(self.ctx._h, self._h, _p(sid), C.c_int32(n_samples), C.byref(h),
                                         C.byref(nu)), "mirge_collapse")
        else:
            w = np.ascontiguousarray(weights, dtype=np.uint32)
            if w.shape[0] != len(self):
                raise ValueError("one weight per read")
            _check(load().mirge_collapse_weighted(self.ctx._h, self._h, _p(sid), C.c_int32(n_samples), _p(w), C.byref(h),
                                                  C.byref(nu)), "mirge_collapse_weighted")
        return DeviceReads(self.ctx, h)

    def counts(self) -> Tuple[np.ndarray, np.ndarray]:
        """(counts [U, S] uint32, first_index [U] int64) of a collapse result."""
        n, S = len(self), self.n_samples
        cnt = np.zeros((n, max(S, 1)), dtype=np.uint32)
        first = np.zeros(n, dtype=np.int64)
        _check(load().mirge_collapse_fetch(self.ctx._h, self._h, _p(cnt), _p(first)), "mirge_collapse_fetch")
        return cnt, first

    def first_appearance_order(self) -> np.ndarray:
        """Indices of the unique reads in the order their first copy appeared in the raw reads (``mirge_collapse_order``)."""
        order = np.zeros(len(self), dtype=np.int64)
        _check(load().mirge_collapse_order(self.ctx._h, self._h, _p(order)), "mirge_collapse_order")
        return order

    def sorted_order(self) -> np.ndarray:
        """Indices of the unique reads in the order of the sorted sequences (Python string order: the index of the reference's
        outer-joined sample matrix, digest.py:243), from a radix sort on the device (``mirge_collapse_order_sorted``)."""
        order = np.zeros(len(self), dtype=np.int64)
        _check(load().mirge_collapse_order_sorted(self.ctx._h, self._h, _p(order)), "mirge_collapse_order_sorted")
        return order

    def range_sample(self, k: int = 256) -> np.ndarray:
        """k evenly spaced quantiles of the dictionary's first-word sort keys (``mirge_reads_range_sample``): what a rank puts
        into the pool the sharded run's range splitters are chosen from (``multigpu.choose_splitters``)"""
        out = np.zeros(int(k), dtype=np.uint64)
        _check(load().mirge_reads_range_sample(self.ctx._h, self._h, C.c_int32(int(k)), _p(out)), "mirge_reads_range_sample")
        return out

    def range_split(self, splitters: np.ndarray):
        """The dictionary ordered by (owner range, handle index) -> (FlatSeqs, counts [U, S] uint32, bounds int64 [n_parts + 1]);
        range q = keys in [splitters[q-1], splitters[q]) = rows bounds[q] .. bounds[q+1] (``mirge_reads_range_split``)."""
        sp = np.ascontiguousarray(splitters, dtype=np.uint64)
        n_parts = int(sp.shape[0]) + 1
        n, S = len(self), max(self.n_samples, 1)
        off = np.zeros(n + 1, dtype=np.int64)
        data = np.empty(max(load().mirge_reads_total_bases(self._h), 1), dtype=np.uint8)
        cnt = np.zeros((n, S), dtype=np.uint32)
        bounds = np.zeros(n_parts + 1, dtype=np.int64)
        _check(load().mirge_reads_range_split(self.ctx._h, self._h, _p(sp) if sp.size else C.c_void_p(0), C.c_int32(n_parts), _p(data),
                                              _p(off), _p(cnt) if cnt.size else C.c_void_p(0), _p(bounds)), "mirge_reads_range_split")
        return FlatSeqs(data[:off[-1]], off), cnt, bounds

    def nonzero_per_sample(self) -> np.ndarray:
        """unique reads with a count, per sample column (``mirge_collapse_nonzero``)"""
        out = np.zeros(max(self.n_samples, 1), dtype=np.int64)
        _check(load().mirge_collapse_nonzero(self.ctx._h, self._h, _p(out)), "mirge_collapse_nonzero")
        return out

    def set_counts(self, counts: np.ndarray):
        counts = np.ascontiguousarray(counts, dtype=np.uint32).reshape(len(self), -1)
        _check(load().mirge_reads_set_counts(self.ctx._h, self._h, _p(counts), C.c_int32(counts.shape[1])),
               "mirge_reads_set_counts")

    def close(self):
        if self._h:
            load().mirge_reads_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CascadeResult:
    """Device-resident per-read annotation (``mirge_result``)."""

    def __init__(self, ctx: Context, handle: C.c_void_p, reads: DeviceReads, n_pass: int):
        self.ctx, self._h, self.reads, self.n_pass = ctx, handle, reads, n_pass
        _track(self)

    def fetch(self):
        """(pass int8, ref int32, off int32, mm int8) in the read set's order; -1 = unannotated."""
        n = len(self.reads)
        ps = np.empty(n, dtype=np.int8)
        ref = np.empty(n, dtype=np.int32)
        off = np.empty(n, dtype=np.int32)
        mm = np.empty(n, dtype=np.int8)
        _check(load().mirge_result_fetch(self.ctx._h, self._h, _p(ps), _p(ref), _p(off), _p(mm)),
               "mirge_result_fetch")
        return ps, ref, off, mm

    def close(self):
        if self._h:
            load().mirge_result_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cascade_args(libs: Sequence[Optional[DeviceLibrary]], policies: Sequence[MirgePolicy]):
    """The two C arrays ``mirge_cascade_run`` takes, built once by callers that run many read sets through one
    library set (this sits on the host's critical path between the collapse's sync and the first pass)."""
    n_pass = len(policies)
    arr = (C.c_void_p * n_pass)(*[(lb._h if lb is not None else C.c_void_p(0)) for lb in libs])
    pol = (MirgePolicy * n_pass)(*policies)
    return arr, pol, C.c_int32(n_pass)


def cascade_run(ctx: Context, reads: DeviceReads, libs: Sequence[Optional[DeviceLibrary]],
                policies: Sequence[MirgePolicy], prepared=None) -> CascadeResult:
    arr, pol, n_pass = prepared if prepared is not None else cascade_args(libs, policies)
    h = C.c_void_p()
    _check(load().mirge_cascade_run(ctx._h, reads._h, arr, pol, n_pass, C.byref(h)), "mirge_cascade_run")
    return CascadeResult(ctx, h, reads, n_pass.value)


def cascade_prepare(ctx: Context, reads: DeviceReads, libs: Sequence[Optional[DeviceLibrary]],
                    policies: Sequence[MirgePolicy], prepared=None) -> None:
    """Build the probe / plan tables a cascade over ``reads`` needs now (``mirge_cascade_prepare``) and wait for them."""
    arr, pol, n_pass = prepared if prepared is not None else cascade_args(libs, policies)
    _check(load().mirge_cascade_prepare(ctx._h, reads._h, arr, pol, n_pass), "mirge_cascade_prepare")


def cascade_walks(ctx: "Context"):
    """(walks over the bulk one-word group's list, passes answered by a whole-read lookup inside another pass's walk, passes) of
    the ctx's current cascade configuration (``mirge_cascade_walks``)."""
    w = (C.c_int32 * 3)()
    _check(load().mirge_cascade_walks(ctx._h, w), "mirge_cascade_walks")
    return int(w[0]), int(w[1]), int(w[2])


def cascade_wg_times(ctx: "Context"):
    """milliseconds every workgroup of the last profiled bulk-cascade launch took (``mirge_cascade_wg_times``), or None"""
    g, khz = C.c_int32(0), C.c_int32(0)
    _check(load().mirge_cascade_wg_times(ctx._h, None, 0, C.byref(g), C.byref(khz)), "mirge_cascade_wg_times")
    if g.value <= 0:
        return None
    t = np.zeros(g.value, dtype=np.uint32)
    _check(load().mirge_cascade_wg_times(ctx._h, _p(t), C.c_int32(g.value), C.byref(g), C.byref(khz)), "mirge_cascade_wg_times")
    return t.astype(np.float64) / max(khz.value, 1)


def collapse_cascade(ctx: Context, raw: DeviceReads, libs: Sequence[Optional[DeviceLibrary]],
                     policies: Sequence[MirgePolicy], prepared=None) -> Tuple[DeviceReads, CascadeResult]:
    """One sample: collapse + cascade in one call (``mirge_collapse_cascade``) -> (unique reads, annotation)."""
    arr, pol, n_pass = prepared if prepared is not None else cascade_args(libs, policies)
    hu, hr = C.c_void_p(), C.c_void_p()
    nu = C.c_int64()
    _check(load().mirge_collapse_cascade(ctx._h, raw._h, arr, pol, n_pass, C.byref(hu), C.byref(nu), C.byref(hr)),
           "mirge_collapse_cascade")
    uniq = DeviceReads(ctx, hu)
    return uniq, CascadeResult(ctx, hr, uniq, n_pass.value)


def count_join(ctx: Context, uniq: DeviceReads, res: CascadeResult, exact_pass: int, iso_pass: int,
               n_mirna: int):
    """-> (class_sums [n_pass, S], exact [n_mirna, S], iso [n_mirna, S]) int64."""
    S = uniq.n_samples
    cls = np.zeros((res.n_pass, S), dtype=np.int64)
    ex = np.zeros((max(n_mirna, 1), S), dtype=np.int64)
    iso = np.zeros((max(n_mirna, 1), S), dtype=np.int64)
    _check(load().mirge_count_join(ctx._h, uniq._h, res._h, C.c_int32(exact_pass), C.c_int32(iso_pass),
                                   C.c_int64(n_mirna), _p(cls), _p(ex), _p(iso)), "mirge_count_join")
    return cls, ex[:n_mirna], iso[:n_mirna]


def count_join_host(ctx: Context, ps: np.ndarray, ref: np.ndarray, counts: np.ndarray, n_pass: int,
                    exact_pass: int, iso_pass: int, n_mirna: int):
    """``count_join`` from host arrays (pass int8 [n], ref int32 [n], counts uint32 [n, S])."""
    ps = np.ascontiguousarray(ps, dtype=np.int8)
    ref = np.ascontiguousarray(ref, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32).reshape(ps.shape[0], -1)
    S = counts.shape[1]
    cls = np.zeros((n_pass, S), dtype=np.int64)
    ex = np.zeros((max(n_mirna, 1), S), dtype=np.int64)
    iso = np.zeros((max(n_mirna, 1), S), dtype=np.int64)
    _check(load().mirge_count_join_host(ctx._h, _p(ps), _p(ref), _p(counts), C.c_int64(ps.shape[0]), C.c_int32(S),
                                        C.c_int32(n_pass), C.c_int32(exact_pass), C.c_int32(iso_pass),
                                        C.c_int64(n_mirna), _p(cls), _p(ex), _p(iso)), "mirge_count_join_host")
    return cls, ex[:n_mirna], iso[:n_mirna]


TALLY_POSITIONS = 32


def variant_tally(ctx: Context, uniq: DeviceReads, res: CascadeResult, exact_pass: int, iso_pass: int,
                  fam_of_ref: np.ndarray, targets: FlatSeqs, retained: Optional[np.ndarray], freq: np.ndarray,
                  per_read: bool = True):
    """``mirge_variant_tally`` (config 5 / rows a16, N1) -> dict: n_seqs, seq_true, count_true, canon, kept_exact
    [n_fam, S]; census [n_fam, 32, 4, 4, 3, S]; diag, state [n reads] (handle order) when ``per_read``."""
    n_fam, S, n = len(targets), uniq.n_samples, len(uniq)
    fam_of_ref = np.ascontiguousarray(fam_of_ref, dtype=np.int32)
    tdata = np.ascontiguousarray(targets.data, dtype=np.uint8)
    toff = np.ascontiguousarray(targets.offsets, dtype=np.int64)
    freq = np.ascontiguousarray(freq, dtype=np.float64)
    assert freq.shape[0] == S
    ret = None if retained is None else np.ascontiguousarray(retained, dtype=np.uint8)
    assert ret is None or ret.shape[0] == n
    tabs = np.empty((5, max(n_fam, 1), S), dtype=np.int64)  # every cell is written by the call
    cen = np.empty((max(n_fam, 1), TALLY_POSITIONS, 4, 4, 3, S), dtype=np.int64)
    diag = np.empty(max(n, 1), dtype=np.int8) if per_read else None
    state = np.empty(max(n, 1), dtype=np.int8) if per_read else None
    if n_fam == 0:
        tabs = np.zeros((5, 0, S), dtype=np.int64)
    _check(load().mirge_variant_tally(ctx._h, uniq._h, res._h, C.c_int32(exact_pass), C.c_int32(iso_pass), _p(fam_of_ref),
                                      C.c_int64(fam_of_ref.shape[0]), _p(tdata) if tdata.size else C.c_void_p(0), _p(toff),
                                      C.c_int64(n_fam), _p(ret), _p(freq), _p(tabs) if n_fam else _p(np.zeros(1, np.int64)),
                                      _p(cen), _p(diag), _p(state)), "mirge_variant_tally")
    out = dict(zip(("n_seqs", "seq_true", "count_true", "canon", "kept_exact"), tabs[:, :n_fam]))
    out["census"] = cen[:n_fam]
    if per_read:
        out["diag"], out["state"] = diag[:n], state[:n]
    return out


def annotation_csv(mapped_path, unmapped_path, header: str, seqs: FlatSeqs, ps: np.ndarray, ref: np.ndarray,
                   counts: np.ndarray, rows: np.ndarray, col_of_pass: Sequence[int], n_name_cols: int,
                   names_by_pass: Sequence[Optional[FlatSeqs]]):
    """``mapped.csv`` / ``unmapped.csv`` (mirge/__main__.py:164-173) from flat arrays; ``names_by_pass[p]`` = the
    reference names of pass p's library as a FlatSeqs (None: the pass has none)."""
    n_pass = len(col_of_pass)
    data = np.ascontiguousarray(seqs.data, dtype=np.uint8)
    off = np.ascontiguousarray(seqs.offsets, dtype=np.int64)
    ps = np.ascontiguousarray(ps, dtype=np.int8)
    ref = np.ascontiguousarray(ref, dtype=np.int32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32).reshape(ps.shape[0], -1)
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    col = (C.c_int32 * n_pass)(*[int(x) for x in col_of_pass])
    keep = []
    nd, no, nn = (C.c_void_p * n_pass)(), (C.c_void_p * n_pass)(), (C.c_int64 * n_pass)()
    for p in range(n_pass):
        fs = names_by_pass[p]
        if fs is None:
            nd[p], no[p], nn[p] = None, None, 0
            continue
        d = np.ascontiguousarray(fs.data, dtype=np.uint8)
        o = np.ascontiguousarray(fs.offsets, dtype=np.int64)
        keep += [d, o]
        nd[p], no[p], nn[p] = d.ctypes.data if d.size else None, o.ctypes.data, len(fs)
    enc = lambda x: None if x is None else str(x).encode()
    _check(load().mirge_annotation_csv(enc(mapped_path), enc(unmapped_path), header.encode(), _p(data) if data.size else C.c_void_p(0),
                                       _p(off), _p(ps), _p(ref), _p(counts), C.c_int32(counts.shape[1]), _p(rows),
                                       C.c_int64(rows.shape[0]), C.c_int32(n_pass), col, C.c_int32(n_name_cols), nd, no, nn),
           "mirge_annotation_csv")


def _name_tables(col_of_pass, names_by_pass):
    """ctypes views of the per-pass reference-name tables the CSV formatters take (+ the arrays that must outlive the call)"""
    n_pass = len(col_of_pass)
    col = (C.c_int32 * n_pass)(*[int(x) for x in col_of_pass])
    keep = []
    nd, no, nn = (C.c_void_p * n_pass)(), (C.c_void_p * n_pass)(), (C.c_int64 * n_pass)()
    for p in range(n_pass):
        fs = names_by_pass[p]
        if fs is None:
            nd[p], no[p], nn[p] = None, None, 0
            continue
        d = np.ascontiguousarray(fs.data, dtype=np.uint8)
        o = np.ascontiguousarray(fs.offsets, dtype=np.int64)
        keep += [d, o]
        nd[p], no[p], nn[p] = d.ctypes.data if d.size else None, o.ctypes.data, len(fs)
    return n_pass, col, nd, no, nn, keep


def annotation_csv_device_sizes(ctx: "Context", uniq: "DeviceReads", res: "CascadeResult", rows: np.ndarray, col_of_pass: Sequence[int],
                                n_name_cols: int, names_by_pass: Sequence[Optional[FlatSeqs]]):
    """(bytes in mapped.csv, bytes in unmapped.csv) the listed rows take (``mirge_annotation_csv_device_sizes``): a rank's stretch of
    a sharded run's files; None when a reference name needs CSV quoting (the run then takes rank 0's host route)."""
    n_pass, col, nd, no, nn, keep = _name_tables(col_of_pass, names_by_pass)
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    out = np.zeros(2, dtype=np.int64)
    rc = load().mirge_annotation_csv_device_sizes(ctx._h, uniq._h, res._h, _p(rows) if rows.size else C.c_void_p(0), C.c_int64(rows.shape[0]),
                                                  C.c_int32(n_pass), col, C.c_int32(n_name_cols), nd, no, nn, _p(out))
    if rc == -4:
        return None
    _check(rc, "mirge_annotation_csv_device_sizes")
    return int(out[0]), int(out[1])


def annotation_csv_device_at(ctx: "Context", uniq: "DeviceReads", res: "CascadeResult", mapped_path, unmapped_path, mapped_off: int,
                             unmapped_off: int, rows: np.ndarray, col_of_pass: Sequence[int], n_name_cols: int,
                             names_by_pass: Sequence[Optional[FlatSeqs]]):
    """The listed rows' text at the given byte offsets of the two EXISTING files (``mirge_annotation_csv_device_at``)."""
    n_pass, col, nd, no, nn, keep = _name_tables(col_of_pass, names_by_pass)
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    _check(load().mirge_annotation_csv_device_at(ctx._h, uniq._h, res._h, str(mapped_path).encode(), str(unmapped_path).encode(),
                                                 C.c_int64(int(mapped_off)), C.c_int64(int(unmapped_off)),
                                                 _p(rows) if rows.size else C.c_void_p(0), C.c_int64(rows.shape[0]), C.c_int32(n_pass), col,
                                                 C.c_int32(n_name_cols), nd, no, nn), "mirge_annotation_csv_device_at")


def annotation_csv_device(ctx: "Context", uniq: "DeviceReads", res: "CascadeResult", mapped_path, unmapped_path, header: str,
                          rows: np.ndarray, col_of_pass: Sequence[int], n_name_cols: int,
                          names_by_pass: Sequence[Optional[FlatSeqs]]) -> bool:
    """``mapped.csv`` / ``unmapped.csv`` formatted on the GPU from the device-resident reads, counts and annotation
    (``mirge_annotation_csv_device``).  False: a reference name needs CSV quoting -- call ``annotation_csv`` instead."""
    n_pass, col, nd, no, nn, keep = _name_tables(col_of_pass, names_by_pass)
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    enc = lambda x: None if x is None else str(x).encode()
    rc = load().mirge_annotation_csv_device(ctx._h, uniq._h, res._h, enc(mapped_path), enc(unmapped_path), header.encode(),
                                            _p(rows) if rows.size else C.c_void_p(0), C.c_int64(rows.shape[0]), C.c_int32(n_pass),
                                            col, C.c_int32(n_name_cols), nd, no, nn)
    if rc == -4:
        return False
    _check(rc, "mirge_annotation_csv_device")
    return True
