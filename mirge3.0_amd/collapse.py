"""Read collapse and the sample matrix -- host mirror of ``mirge/libs/digest.py:105-302``.

``baking(args, inFileArray, inFileBaseArray, workDir)`` keeps the reference's signature and
return value ``(DataFrame, sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique)``.
The per-chunk dict counting (``cutadapt()``, digest.py:320-375), the dict merge (:141-163) and
the pandas outer join of the samples (:237-245) are ONE ``mirge_collapse`` call on the GPU: all
samples' reads go in together with a sample id and come back as the U distinct sequences plus a
U x S count matrix.

The cutadapt modifier chain of ``stipulate`` (digest.py:59-101) runs on the GPU too (``trim_from_args`` ->
``mirge_reads_parse_trim``: quality / NextSeq trimming, one 3' or one 5' adapter, N ends, cuts); what is not covered
(more than two adapters, ``--action mask`` / ``lowercase``) raises instead of being skipped.

UMI handling (digest.py:164-205,305-315,334-365) is part of the same device-resident parse
(``mirge_reads_parse_umi``): ``-umi f,b`` slices f bases off the front and b off the back of every counted read
(counts add up); with ``-udd`` the UMI-tagged reads are collapsed first, ``<sample>_umiCounts.csv`` is written from that
result and ONE insert per distinct tagged read goes on, so a count is a number of molecules; ``--qiagenumi`` takes the
UMI from behind the 3' adapter of the untrimmed read (the reference's ``currentSeq.split(trimmed)[1]`` string rule).
Both command lines of the reference's quick start (``docs/source/quick_start.md:282-314``) run this way.
``-tcf`` writes ``<sample>.trim.collapse.fa`` (digest.py:219-229).
"""
from __future__ import annotations

import gzip
import time
from pathlib import Path
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import PASS_COLUMNS, _ffi
from .seqio import FlatSeqs


def read_fastq_sequences(path: str) -> FlatSeqs:
    """Sequence lines of a FASTQ (or FASTA, or one-sequence-per-line) file, plain or .gz."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as fh:
        buf = np.frombuffer(fh.read(), dtype=np.uint8)
    if buf.size == 0:
        return FlatSeqs(np.zeros(0, np.uint8), np.zeros(1, np.int64))
    nl = np.flatnonzero(buf == 10)
    if buf[-1] != 10:
        nl = np.append(nl, buf.size)
    starts = np.concatenate(([0], nl[:-1] + 1))
    ends = nl.copy()
    # strip \r
    cr = (ends > starts) & (buf[np.maximum(ends - 1, 0)] == 13)
    ends = ends - cr
    first = buf[0]
    if first == ord("@"):
        sel = slice(1, None, 4)
    elif first == ord(">"):
        sel = slice(1, None, 2)
    else:
        sel = slice(0, None, 1)
    s, e = starts[sel], ends[sel]
    lens = (e - s).astype(np.int64)
    offsets = np.zeros(lens.shape[0] + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    rows = np.repeat(np.arange(lens.shape[0], dtype=np.int64), lens)
    within = np.arange(int(offsets[-1]), dtype=np.int64) - offsets[:-1][rows]
    return FlatSeqs(buf[s[rows] + within], offsets)


import os as _os
GZ_PIECE_BYTES = int(_os.environ.get("MIRGE_GZ_PIECE_BYTES", 8 << 20))   # text handed to the parser per piece of a streamed .fastq.gz
# a text of this size or more is parsed in parts of this size (mirge_reads_parse takes less than 8 GiB at a time; tests lower it)
TEXT_PIECE_BYTES = int(_os.environ.get("MIRGE_TEXT_PIECE_BYTES", str(2 << 30)))
GZ_AHEAD_PIECE_BYTES = int(_os.environ.get("MIRGE_GZ_AHEAD_PIECE_BYTES", str(48 << 20)))  # text parsed at a time beside a parallel inflation
GZ_QUEUE_DEPTH = 4         # inflated pieces waiting for the GPU (bounds the memory of a stream: ~ depth x piece)
GZ_LOG: List[dict] = []    # one entry per .gz file inflated by mirge_gz_inflate (read_text): what a run reports as its input stage


class GzipRecordStream:
    """A ``.fastq.gz`` as pieces of text, each a whole number of 4-line records, inflated on a worker thread (zlib releases
    the GIL) while the caller uploads and parses the piece before: what ``xopen`` + ``dnaio.read_chunks`` are to the
    reference's worker pool (digest.py:136-140).  One inflate stream runs at 0.3-0.4 GB/s of text, twenty times slower than
    everything behind it, so the sample costs what its inflation costs and no more; the whole text is never held.
    Multi-member files (bgzip, pigz, cat of several .gz) are read through.  A text that does not start with '@' (FASTA, bare
    sequences) is gathered whole instead (``whole_text``): those parsers want to see all of it."""

    def __init__(self, path: str, piece_bytes: int = None, depth: int = None):
        import queue
        import threading
        self.path = str(path)
        self.piece_bytes = int(piece_bytes or GZ_PIECE_BYTES)
        self.q: "queue.Queue" = queue.Queue(maxsize=int(depth or GZ_QUEUE_DEPTH))
        self.inflate_s = 0.0       # the worker's time inside zlib + cutting (its wall time minus waits for the consumer)
        self.text_bytes = 0
        self.compressed_bytes = 0
        self.pieces = 0
        self._err = None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._work, name="mirge-inflate", daemon=True)
        self._thread.start()

    @staticmethod
    def _cut(block: bytes) -> int:
        """length of the longest prefix of `block` that is a whole number of 4-line records (0: none)"""
        n = block.count(b"\n")
        idx = block.rfind(b"\n")
        for _ in range(n % 4):
            idx = block.rfind(b"\n", 0, idx)
        return idx + 1

    def _inflated(self, fh):
        """the file's text as blocks of at most a piece, member after member; (block, seconds inside zlib)"""
        import zlib
        d, fresh, buf, first = zlib.decompressobj(31), True, b"", True
        while True:
            if not buf:
                buf = fh.read(1 << 20)
                self.compressed_bytes += len(buf)
                if not buf:
                    if not fresh:  # the file ends inside a member: whatever zlib still holds, then the verdict
                        t0 = time.perf_counter()
                        out = d.flush()
                        if out:
                            yield out, time.perf_counter() - t0
                        if not d.eof:
                            raise EOFError(f"{self.path}: compressed file ended before the end-of-stream marker was reached")
                    return
            if fresh and not first:
                # BETWEEN members zeros are padding (tape blocks, bgzip's tools), as Python's gzip module and xopen read them
                # -- however many reads of the file they span (a 1 MiB read that ended inside the padding used to hand the
                # rest of it to a new decompressor: 'incorrect header check')
                buf = buf.lstrip(b"\0")
                if not buf:
                    continue
            t0 = time.perf_counter()
            out = d.decompress(buf, self.piece_bytes)  # at most a piece at a time, whatever the ratio
            fresh = first = False
            if d.eof:  # end of a member: the next one, if any, starts in what is left
                buf = d.unused_data
                d, fresh = zlib.decompressobj(31), True
            else:
                buf = d.unconsumed_tail  # (output still pending inside zlib comes out with the next input, or the final flush)
            yield out, time.perf_counter() - t0

    def _put(self, item) -> bool:
        import queue
        while not self._stop.is_set():
            try:
                self.q.put(item, timeout=0.2)  # blocks while the consumer is GZ_QUEUE_DEPTH pieces behind
                return True
            except queue.Full:
                continue
        return False

    def _work(self):
        t_busy = 0.0
        try:
            parts, size, carry, fastq = [], 0, b"", None
            with open(self.path, "rb") as fh:
                for out, dt in self._inflated(fh):
                    t_busy += dt
                    if not out:
                        continue
                    if fastq is None:
                        fastq = out[:1] == b"@"
                    parts.append(out)
                    size += len(out)
                    if fastq and size >= self.piece_bytes:
                        t0 = time.perf_counter()
                        block = carry + b"".join(parts)
                        cut = self._cut(block)
                        piece, carry = block[:cut], block[cut:]
                        parts, size = [], 0
                        t_busy += time.perf_counter() - t0
                        if piece:
                            self.text_bytes += len(piece)
                            self.pieces += 1
                            if not self._put(piece):
                                return
            last = carry + b"".join(parts)
            if last:
                self.text_bytes += len(last)
                self.pieces += 1
                self._put(("whole", last) if not fastq else last)
        except Exception as e:  # noqa: BLE001 -- handed to the consumer
            self._err = e
        finally:
            self.inflate_s = t_busy
            self._stop_put_none()

    def _stop_put_none(self):
        import queue
        while True:
            try:
                self.q.put(None, timeout=0.2)
                return
            except queue.Full:
                if self._stop.is_set():  # nobody is listening: make room
                    try:
                        self.q.get_nowait()
                    except queue.Empty:
                        pass

    def close(self):
        """stop the worker (a consumer that gives up early calls this; harmless after the end)"""
        self._stop.set()
        self._thread.join(timeout=5)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                self._thread.join()
                if self._err is not None:
                    raise self._err
                return
            yield item

    def whole_text(self) -> bytes:
        """everything as one text (a stream somebody needs whole: UMI handling, FASTA)"""
        out = []
        for item in self:
            out.append(item[1] if isinstance(item, tuple) else item)
        data = b"".join(out)
        return unwrap_fasta(data) if data[:1] == b">" else data


class TextRecordStream:
    """A FASTQ text already in host memory (a memory-mapped file, an inflated ``.gz``) as pieces of whole 4-line records: what a
    text too large for ONE ``mirge_reads_parse`` call -- 8 GiB, some 150 M reads -- is passed as.  Same protocol as
    ``GzipRecordStream`` (iterate, ``close``, ``whole_text``, the timing fields)."""

    def __init__(self, data, piece_bytes: int = None):
        self.data = data if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
        self.piece_bytes = int(piece_bytes or TEXT_PIECE_BYTES)
        self.inflate_s = 0.0
        self.text_bytes = 0
        self.pieces = 0

    def _newlines(self, a: int, b: int) -> int:
        """line ends in data[a:b], counted 64 MB at a time on a few threads (numpy releases the GIL inside the comparison)"""
        from concurrent.futures import ThreadPoolExecutor
        step = 64 << 20
        spans = [(x, min(b, x + step)) for x in range(a, b, step)]
        if len(spans) <= 1:
            return int(np.count_nonzero(self.data[a:b] == 10))
        with ThreadPoolExecutor(max_workers=min(8, len(spans))) as pool:
            return int(sum(pool.map(lambda ab: int(np.count_nonzero(self.data[ab[0]:ab[1]] == 10)), spans)))

    def _cut(self, a: int, b: int) -> int:
        """the largest e <= b such that data[a:e] is a whole number of 4-line records (a: none fits)"""
        drop = self._newlines(a, b) % 4  # lines to give back, besides the partial one at the end
        e = b
        for _ in range(drop + 1):
            # the last line end in front of e (searched in growing windows from the back: lines are short)
            w, found = 1 << 16, -1
            while found < 0 and e > a:
                lo = max(a, e - w)
                hits = np.flatnonzero(self.data[lo:e] == 10)
                if hits.size:
                    found = lo + int(hits[-1])
                elif lo == a:
                    return a
                w *= 16
            if found < 0:
                return a
            e = found  # exclusive: the next search looks in front of this line end
        return e + 1

    def __iter__(self):
        n, at = int(self.data.size), 0
        while at < n:
            end = min(n, at + self.piece_bytes)
            if end < n:
                end = self._cut(at, end)
                if end <= at:
                    raise RuntimeError(f"a FASTQ record longer than {self.piece_bytes} bytes: not a text this parser takes in parts")
            self.text_bytes += end - at
            self.pieces += 1
            yield self.data[at:end]
            at = end

    def close(self):
        pass

    def whole_text(self):
        return self.data


class GzRouteDeclined(Exception):
    """mirge_gz_inflate did not take the file (or its CRC-32 did not match): what was parsed ahead is dropped, zlib takes over"""


def record_cut(data: np.ndarray, at: int, end: int) -> int:
    """The largest e in (at, end] where a FASTQ record starts (data[at] starts one), found from the text alone: a line that starts
    with '@' and whose second successor starts with '+' is a header -- a quality line may start with '@', but two lines behind a
    quality line stands a sequence, and sequences are letters.  ``at`` when no such line lies in data[at:end]."""
    w = 1 << 18
    while True:
        lo = max(at, end - w)
        starts = np.flatnonzero(data[lo:end] == 10) + (lo + 1)
        if lo == at:
            starts = np.concatenate(([at], starts))
        starts = starts[starts < end]
        if starts.size >= 3:
            ok = (data[starts[:-2]] == 64) & (data[starts[2:]] == 43)
            hit = np.flatnonzero(ok)
            if hit.size and int(starts[hit[-1]]) > at:
                return int(starts[hit[-1]])
        if lo == at:
            return at
        w *= 8


class ParallelGzipStream:
    """A ``.fastq.gz`` inflated on all host cores (``mirge_gz_inflate_progress`` on a thread of its own) WHILE the caller uploads
    and parses the part of the text that is already final: pieces of whole records, cut where ``record_cut`` finds a header.
    The file's CRC-32 is known at the end only -- a mismatch, or a file the parallel route does not take, raises
    ``GzRouteDeclined`` from the iteration and the caller starts over with ``GzipRecordStream``.  Same protocol as that class."""

    def __init__(self, path: str, piece_bytes: int = None, reserved: int = 0):
        self.path = str(path)
        self.piece_bytes = int(piece_bytes or GZ_AHEAD_PIECE_BYTES)
        self.raw = np.memmap(self.path, dtype=np.uint8, mode="r")  # the compressed bytes straight from the page cache
        self.compressed_bytes = int(self.raw.size)
        self.inflate_s = 0.0
        self.text_bytes = 0
        self.pieces = 0
        self._reserved = int(reserved)  # host memory promised by _gz_reserve: given back by close()
        self._t0 = time.perf_counter()
        try:
            # (the cores are shared by the inflations admitted side by side)
            self.job = _ffi.GzInflation(self.raw, threads=max(1, _ffi.gz_threads() // max(1, _gz_in_flight)))
        except BaseException:
            _gz_release(self._reserved)
            self._reserved = 0
            raise
        self._logged = False
        self._whole_handed_out = False

    def _finished(self) -> bool:
        """the inflation has ended well (raises when it has ended otherwise)"""
        ok = self.job.wait()
        if not self._logged:
            self._logged = True
            self.inflate_s = time.perf_counter() - self._t0  # an upper bound when the consumer came late
            if ok:
                GZ_LOG.append({"path": self.path, "gz_MB": round(self.compressed_bytes / 1e6, 1), "text_MB": round(self.job.text().size / 1e6, 1),
                               "parallel_inflate_s": round(self.inflate_s, 4), "parsed_beside": True})
        if not ok:
            raise GzRouteDeclined(self.path)
        return True

    def __iter__(self):
        at, total = 0, None
        out = self.job.out
        while True:
            if total is None and not self.job.running():
                self._finished()
                total = int(self.job.text().size)
            done = total if total is not None else self.job.done()
            if at == 0 and done > 0 and out[0] != 64:  # not FASTQ: those parsers want the whole text
                self._finished()
                self.text_bytes, self.pieces = int(self.job.text().size), 1
                yield ("whole", bytes(self.job.text()))
                return
            if total is not None and at >= total:
                return
            want = self.piece_bytes if total is None else 1
            if done - at >= want:
                end = min(done, at + TEXT_PIECE_BYTES)
                if total is None or end < total:
                    end = record_cut(out, at, end)
                if end > at:
                    self.text_bytes += end - at
                    self.pieces += 1
                    yield out[at:end]
                    at = end
                    continue
                if total is not None:
                    raise RuntimeError(f"{self.path}: no FASTQ record starts in {TEXT_PIECE_BYTES} bytes of text")
            time.sleep(0.0003)

    def close(self):
        """the consumer is done with every piece (they were uploaded): the text buffer goes back for the next sample"""
        self.job.wait()  # (the buffers must outlive the worker)
        if not self._whole_handed_out:
            self.job.release()
        need, self._reserved = self._reserved, 0
        _gz_release(need)

    def __del__(self):  # a stream nobody iterated (an exception upstream) must not hold its reservation for ever
        try:
            need, self._reserved = getattr(self, "_reserved", 0), 0
            _gz_release(need)
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def whole_text(self):
        if not self.job.wait():
            return GzipRecordStream(self.path).whole_text()
        self._finished()
        t = self.job.text()
        self._whole_handed_out = True  # the caller keeps a view: this buffer is not reused
        return unwrap_fasta(bytes(t)) if bytes(t[:1]) == b">" else t


def _host_memory_available() -> int:
    """bytes of host memory this process may still take: MemAvailable, or what is left of the cgroup's limit when that is less"""
    avail = 1 << 62
    try:
        with open("/proc/meminfo") as fh:
            for ln in fh:
                if ln.startswith("MemAvailable"):
                    avail = int(ln.split()[1]) * 1024
                    break
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v != "max":
                avail = min(avail, int(v) - int(open(cur).read().strip()))
        except (OSError, ValueError):
            pass
    return avail


import threading as _threading
_gz_admit_lock = _threading.Lock()
_gz_reserved = 0        # host bytes promised to the parallel inflations in flight (read_texts starts up to four at once)
_gz_in_flight = 0
GZ_MAX_IN_FLIGHT = int(_os.environ.get("MIRGE_GZ_MAX_IN_FLIGHT", "2"))


def _gz_host_bytes(path: str) -> int:
    """What one parallel inflation of this file may take from the host: the text (ISIZE of a single member -- the real size
    modulo 2^32, so a 343:1 file of repeats is seen for what it is --, at least eight times the compressed size: FASTQ deflates
    4-6 x and a file of several members shows the last member's size only) plus the 2-byte symbols the chunks decode into."""
    size = _os.path.getsize(path)
    isize = 0
    try:
        with open(path, "rb") as fh:
            fh.seek(max(0, size - 4))
            isize = int.from_bytes(fh.read(4), "little")
    except OSError:
        pass
    return 3 * max(isize, 8 * size)


def _gz_reserve(path: str) -> int:
    """Admission of a parallel inflation: its memory is RESERVED under a lock before it starts, against what the host has left
    minus what the inflations in flight were promised (four files admitted by four threads at the same moment, each looking at
    the same MemAvailable before any had touched a page, could together ask for twice the host's memory: round 4's review).
    At most GZ_MAX_IN_FLIGHT at a time.  Returns the bytes reserved (hand them to ``_gz_release``), 0 = take the streamed route."""
    global _gz_reserved, _gz_in_flight
    need = _gz_host_bytes(path)
    with _gz_admit_lock:
        if _gz_in_flight >= GZ_MAX_IN_FLIGHT or need >= (_host_memory_available() - _gz_reserved) // 2:
            return 0
        _gz_reserved += need
        _gz_in_flight += 1
        return need


def _gz_release(need: int) -> None:
    global _gz_reserved, _gz_in_flight
    if need:
        with _gz_admit_lock:
            _gz_reserved -= need
            _gz_in_flight -= 1


def _gz_fits_in_memory(path: str) -> bool:
    """The parallel inflater holds the whole text in host memory (the streamed route holds a few pieces): taken only when what it
    needs (``_gz_host_bytes``) is less than half of what the host has left beside the inflations already admitted."""
    with _gz_admit_lock:
        return _gz_host_bytes(path) < (_host_memory_available() - _gz_reserved) // 2


def read_text(path: str, stream: bool = False):
    """The file's bytes for the device-side parser (``mirge_reads_parse``): memory-mapped when plain; a ``.gz`` inflated whole,
    or -- ``stream=True`` -- as a ``GzipRecordStream`` whose worker has already started.  A FASTA whose sequences are wrapped
    over several lines (dnaio reads those) is unwrapped here: the device parser finds records by line number and refuses
    anything else."""
    if str(path).endswith(".gz"):
        # first choice: the whole member inflated on all host cores (mirge_gz_inflate: cut at deflate block starts found by
        # search, verified against the file's CRC-32); it declines what it is not made for -- small files, several ordinary
        # members, anything that is not text -- and those are inflated by zlib, piece by piece beside the parse (stream) or whole
        data = None
        # (up to 8 GiB of compressed data -- the text is held whole in host memory, some 6 x that; beyond, the streamed route's
        # few pieces bound the memory a sample takes)
        sized = (2 << 20) <= _os.path.getsize(path) <= (8 << 30)
        if stream and _os.environ.get("MIRGE_GZ_PARALLEL", "1") not in ("0", "whole") and sized:
            need = _gz_reserve(path)  # its memory, promised under a lock; 0: the streamed route below
            if need:
                return ParallelGzipStream(path, reserved=need)  # inflated on all cores, parsed beside; declines like the call below
        # (MIRGE_GZ_PARALLEL=whole with stream=True lands here too: inflated whole first, then parsed -- the route README.md
        # documents and bench.py's `inflated_whole_then_parsed` leg measures; a stream the branch above declined does as well)
        need = _gz_reserve(path) if (sized and _os.environ.get("MIRGE_GZ_PARALLEL", "1") != "0") else 0
        if need:
            t0 = time.perf_counter()
            raw = np.fromfile(path, dtype=np.uint8)
            try:
                data = _ffi.gz_inflate(raw)
            finally:
                _gz_release(need)
            if data is not None and data.size >= 3 * TEXT_PIECE_BYTES and not (stream and bytes(data[:1]) == b"@"):
                data = None  # too large for one parse call and nobody to take it in parts: as before
            if data is not None:
                GZ_LOG.append({"path": str(path), "gz_MB": round(raw.size / 1e6, 1), "text_MB": round(data.size / 1e6, 1),
                               "parallel_inflate_s": round(time.perf_counter() - t0, 4)})
            del raw
        if data is None:
            if stream:
                return GzipRecordStream(path)
            with gzip.open(path, "rb") as fh:
                data = fh.read()
    else:
        import os
        if os.path.getsize(path) == 0:
            return b""
        data = np.memmap(path, dtype=np.uint8, mode="r")  # the pages go from the page cache to the GPU: no read() copy
    if bytes(data[:1]) == b">":
        data = unwrap_fasta(bytes(data))
    elif stream and len(data) >= 3 * TEXT_PIECE_BYTES and bytes(data[:1]) == b"@":
        # mirge_reads_parse takes less than 8 GiB at a time (it says so itself): a larger FASTQ goes in parts of whole records
        return TextRecordStream(data)
    return data


def read_texts(paths, depth: int = 4, stream: bool = False):
    """``read_text`` of every file, in order, read ahead by up to ``depth`` files on worker threads: gunzipping a sample
    (zlib releases the GIL) then runs beside the previous sample's transfer and parse, and several .gz samples inflate
    on several cores -- a single inflate stream is the slowest stage of a run from compressed FASTQ.  ``stream=True``: a
    ``.gz`` comes as a ``GzipRecordStream`` (opened -- its worker inflating -- up to ``depth`` files ahead)."""
    from concurrent.futures import ThreadPoolExecutor
    paths = [str(p) for p in paths]
    if len(paths) <= 1:
        for p in paths:
            yield read_text(p, stream)
        return
    with ThreadPoolExecutor(max_workers=max(1, min(depth, len(paths)))) as pool:
        pending = [pool.submit(read_text, p, stream) for p in paths[:depth]]
        nxt = len(pending)
        while pending:
            text = pending.pop(0).result()
            if nxt < len(paths):
                pending.append(pool.submit(read_text, paths[nxt], stream))
                nxt += 1
            yield text


def unwrap_fasta(data: bytes) -> bytes:
    """FASTA text -> one sequence line per record.  A newline stays when it ends a header line or when the next line
    is a header (or the text ends); every other newline (and '\\r') sits inside a sequence and is dropped."""
    buf = np.frombuffer(data, dtype=np.uint8)
    nl = np.flatnonzero(buf == 10)
    if nl.size == 0:
        return data
    line_start = np.concatenate(([0], nl[:-1] + 1))
    is_header = buf[np.minimum(line_start, buf.size - 1)] == ord(">")
    nxt = nl + 1
    next_is_header = (nxt >= buf.size) | (buf[np.minimum(nxt, buf.size - 1)] == ord(">"))
    drop_nl = nl[~(is_header | next_is_header)]
    if drop_nl.size == 0:
        return data
    keep = np.ones(buf.size, dtype=bool)
    keep[drop_nl] = False
    cr = drop_nl[(drop_nl > 0)] - 1
    keep[cr[buf[cr] == 13]] = False
    return buf[keep].tobytes()


ILLUMINA_3P = 'TGGAATTCTCGGGTGCCAAGGAACTCCAG'  # what `-a illumina` stands for (mirge/__main__.py:65-83)
ILLUMINA_5P = 'GTTCAGAGTTCTACAGTCCGACGATC'     # what `-g illumina` stands for


def adapters_from_args(args):
    """``args.adapters`` as the reference has it after mirge/__main__.py:65-83: a list of (kind, sequence) in command-line
    order, 'illumina' spelled out -- with ONE adapter by its kind, with TWO by its position (the first is taken for the 3'
    adapter, the second for the 5' one, whatever their flags were: the reference's own rule).  Plain strings (3' adapters)
    and a separate ``front`` list are accepted from callers that build the namespace themselves."""
    adapters = getattr(args, "adapters", None) or []
    if isinstance(adapters, str):
        adapters = [adapters]
    adapters = [tuple(a) if isinstance(a, (tuple, list)) else ("back", a) for a in adapters]
    front = getattr(args, "front", None) or []
    if isinstance(front, str):
        front = [front]
    adapters += [tuple(f) if isinstance(f, (tuple, list)) else ("front", f) for f in front]
    if len(adapters) == 2:
        adapters = [(k, (ILLUMINA_3P, ILLUMINA_5P)[i] if q == "illumina" else q) for i, (k, q) in enumerate(adapters)]
    elif len(adapters) == 1:
        adapters = [(k, {"back": ILLUMINA_3P, "front": ILLUMINA_5P}.get(k, q) if q == "illumina" else q) for k, q in adapters]
    return adapters


def parse_adapter_spec(kind: str, spec: str) -> dict:
    """One adapter specification as cutadapt's parser reads it (the reference hands ``args.adapters`` to
    ``make_adapters_from_specifications`` unchanged, digest.py:66-84; docs/source/quick_start.md:213-220 shows a linked one):
      ``SEQ``            a regular 3' (-a) / 5' (-g) adapter
      ``^SEQ`` (-g)      anchored 5': the whole adapter at the read's first base;  ``SEQ$`` (-a): anchored 3', up to its last base
      ``A...B``          ONE linked adapter: 5' part A, then 3' part B in what follows.  Under ``-a`` A is anchored and required
                         and B optional; under ``-g`` A is a regular 5' adapter and BOTH are required (a read without either
                         stays as it is).  ``^A`` / ``B$`` anchor a part explicitly.
    Refused (NotImplementedError, as before): non-internal forms (``XSEQ`` / ``SEQX``), ``name=`` prefixes, ``;parameters``,
    ``file:`` references, ``-b``.  -> dict(kind, seq, anchored) or dict(linked=True, front, back, front_anchored, back_anchored,
    front_required, back_required)."""
    if kind not in ("back", "front"):
        raise NotImplementedError("only -a (3') and -g (5') adapters are supported (-b 'anywhere' adapters are not)")
    if spec.startswith("file:") or "=" in spec or ";" in spec:
        raise NotImplementedError(f"adapter specification {spec!r}: names, parameters and file: references are not supported")
    if "..." in spec:
        front, back = spec.split("...", 1)
        fa, ba = front.startswith("^"), back.endswith("$")
        front, back = front[1 if fa else 0:], back[:-1] if ba else back
        if not front or not back:
            raise NotImplementedError(f"adapter specification {spec!r}: a linked adapter needs both parts (A...B)")
        if "..." in back or any(c in front + back for c in "^$"):
            raise SystemExit(f"adapter specification {spec!r}: not a valid linked adapter")
        if front.upper().startswith("X") or back.upper().endswith("X"):
            raise NotImplementedError(f"adapter specification {spec!r}: non-internal adapters (X) are not supported")
        if kind == "back":
            return dict(linked=True, front=front, back=back, front_anchored=True, back_anchored=ba, front_required=True, back_required=False)
        return dict(linked=True, front=front, back=back, front_anchored=fa, back_anchored=ba, front_required=True, back_required=True)
    anchored = False
    if kind == "front" and spec.startswith("^"):
        spec, anchored = spec[1:], True
    elif kind == "back" and spec.endswith("$"):
        spec, anchored = spec[:-1], True
    if "^" in spec or "$" in spec:
        raise SystemExit(f"adapter specification {spec!r}: '^' belongs in front of a 5' adapter (-g), '$' behind a 3' adapter (-a)")
    if (kind == "front" and spec.upper().startswith("X")) or (kind == "back" and spec.upper().endswith("X")):
        raise NotImplementedError(f"adapter specification {spec!r}: non-internal adapters (X) are not supported")
    return dict(kind=kind, seq=spec, anchored=anchored)


def trim_from_args(args):
    """The cutadapt modifier chain of ``stipulate`` (digest.py:59-101) as the options of ``mirge_reads_parse_trim``:
    ``-q`` (default "10": quality trimming is ALWAYS in the reference's chain), ``-a`` / ``-g`` (one or two adapters, 3'
    or 5', regular or anchored; with two, a read loses the better match -- AdapterCutter with times = 1 --; or ONE linked
    adapter ``A...B``: ``parse_adapter_spec``), ``-nxt``, ``-NX``, ``-u``,
    ``--overlap``, ``--error-rate``, ``-phr``, ``-n`` (repeat the removal), ``--no-indels``, ``--match-read-wildcards``, ``-N``,
    ``--action none``.  More than two adapters and ``--action mask`` / ``lowercase`` are refused."""
    adapters = adapters_from_args(args)
    if len(adapters) > 2 or any(kind not in ("back", "front") for kind, _ in adapters):
        raise NotImplementedError("up to two adapters are supported (-a / -g, in any combination)")
    specs = [parse_adapter_spec(k, q) for k, q in adapters]
    if any(sp.get("linked") for sp in specs) and len(specs) > 1:
        raise NotImplementedError("a linked adapter (A...B) beside another adapter is not supported")
    q = getattr(args, "quality_cutoff", "10")
    qf, qb = 0, -1
    if q is not None:
        vals = [int(v) for v in str(q).split(",")]
        if len(vals) == 1:
            qf, qb = 0, vals[0]
        elif len(vals) == 2:
            qf, qb = vals
        else:
            raise SystemExit("Expected one value or two values separated by comma for the quality cutoff")
    cut = [int(c) for c in (getattr(args, "cut", None) or [])]
    if len(cut) > 2:
        raise SystemExit("You cannot remove bases from more than two ends.")
    if len(cut) == 2 and cut[0] * cut[1] > 0:
        raise SystemExit("You cannot remove bases from the same end twice.")
    nxt = getattr(args, "nextseq_trim", None)
    base = 64 if int(getattr(args, "phred64", 33) or 33) == 64 else 33
    common = dict(quality_back=qb, quality_front=qf, nextseq=-1 if nxt is None else int(nxt),
                  phred_base=base, min_overlap=int(getattr(args, "overlap", 3)),
                  error_rate=float(getattr(args, "error_rate", 0.12)), trim_n=bool(getattr(args, "trim_n", False)),
                  cut=cut, count_per_modifier=getattr(args, "trim_count", "per-modifier") != "once",
                  times=int(getattr(args, "times", 1) or 1), indels=bool(getattr(args, "indels", True)),
                  read_wildcards=bool(getattr(args, "match_read_wildcards", False)),
                  adapter_wildcards=bool(getattr(args, "match_adapter_wildcards", True)),
                  action=str(getattr(args, "action", "trim") or "trim"))
    if specs and specs[0].get("linked"):
        lk = specs[0]
        return _ffi.MirgeTrim.make(adapter=lk["front"], front=True, anchored=lk["front_anchored"], adapter2=lk["back"], front2=False,
                                   anchored2=lk["back_anchored"], linked=True, front_required=lk["front_required"],
                                   back_required=lk["back_required"], **common)
    a1 = specs[0] if specs else dict(kind=None, seq=None, anchored=False)
    a2 = specs[1] if len(specs) > 1 else dict(kind=None, seq=None, anchored=False)
    return _ffi.MirgeTrim.make(adapter=a1["seq"], front=a1["kind"] == "front", anchored=a1["anchored"],
                               adapter2=a2["seq"], front2=a2["kind"] == "front", anchored2=a2["anchored"], **common)


def unpinned_trim_options(args) -> List[str]:
    """The trimming options in use whose behaviour is restated from cutadapt's documentation and sources WITHOUT a vector of a
    real cutadapt behind it (cutadapt is absent from the image and the pool; the plain 3' adapter + quality chain has the user
    guide's cases as known answers): the run says so in run.log instead of silently counting with them."""
    out = []
    adapters = adapters_from_args(args)
    if len(adapters) == 2:
        out.append("two adapters (the better match is removed)")
    for kind, q in adapters:
        if "..." in q:
            out.append("a linked adapter (A...B)")
        elif q.startswith("^") or q.endswith("$"):
            out.append("an anchored adapter (^A / A$)")
    if int(getattr(args, "times", 1) or 1) > 1:
        out.append("-n / --times")
    if not bool(getattr(args, "indels", True)):
        out.append("--no-indels")
    if bool(getattr(args, "match_read_wildcards", False)):
        out.append("--match-read-wildcards")
    if not bool(getattr(args, "match_adapter_wildcards", True)):
        out.append("-N / --no-match-adapter-wildcards")
    if str(getattr(args, "action", "trim") or "trim") == "none":
        out.append("--action none")
    return out


def filter_min_length(reads: FlatSeqs, min_len: int) -> FlatSeqs:
    keep = np.flatnonzero(reads.lengths >= int(min_len))
    if keep.shape[0] == len(reads):
        return reads
    return reads.take(keep)


def collapse_parsed_samples(ctx: _ffi.Context, parsed) -> "_ffi.DeviceReads":
    """The sample matrix (unique reads of the union x S count columns: the outer join of digest.py:243) of several samples whose
    reads are on the device.  Round 6: every sample is collapsed BY ITSELF (the partitioned path, ~0.3 ms per 10 M reads) and the
    dictionaries are merged on the device (``mirge_collapse_merge``); putting the raw reads of all samples through one table with
    sample ids (``mirge_collapse`` on the concatenation, the general path) took 18-37 ms for four 10 M-read samples -- the hot sequences
    of real samples contend for their cells.  MIRGE_JOINT_COLLAPSE=raw keeps that route (the tests compare the two)."""
    S = len(parsed)
    if _os.environ.get("MIRGE_JOINT_COLLAPSE", "merge") == "raw":
        allr = _ffi.DeviceReads.concat(ctx, parsed)
        sid = np.repeat(np.arange(S, dtype=np.int32), [len(p) for p in parsed])
        uniq = allr.collapse(sid, S)
        allr.close()
        return uniq
    dicts = [p.collapse() for p in parsed]
    try:
        return _ffi.DeviceReads.merge(ctx, dicts)
    finally:
        for d in dicts:
            d.close()


def collapse_samples(ctx: _ffi.Context, samples: Sequence[FlatSeqs]):
    """Host sequences of several samples -> the DeviceReads of their joint collapse (U unique reads + a U x S count
    matrix); the route for callers that hold sequences, not files (``baking`` parses the files' text on the GPU)."""
    S = len(samples)
    if S == 1:
        raw = _ffi.DeviceReads.pack(ctx, samples[0])
        uniq = raw.collapse()
        raw.close()
        return uniq
    packed = [_ffi.DeviceReads.pack(ctx, s) for s in samples]  # (round 6: per-sample collapses, merged on the device)
    try:
        return collapse_parsed_samples(ctx, packed)
    finally:
        for r in packed:
            r.close()


def umi_from_args(args):
    """``-umi f,b`` / ``--qiagenumi`` / ``-udd`` as the options of ``mirge_reads_parse_umi``, or None.  The reference reads
    ``umi.split(",")`` everywhere (digest.py:149,167,344,358): two integers."""
    umi = getattr(args, "uniq_mol_ids", None)
    qia = bool(getattr(args, "qiagenumi", False))
    if not umi:
        if qia:
            raise SystemExit("--qiagenumi requires -umi x,y (mirge/libs/parse.py: '-umi x,y Required')")
        return None
    try:
        f, b = (int(x) for x in str(umi).split(","))
    except ValueError:
        raise SystemExit("-umi expects two comma separated integers, e.g. 4,4 or 0,12")
    return _ffi.MirgeUmi.make(f, b, qiagen=qia, dedup=bool(getattr(args, "umiDedup", False)))


def write_umi_counts(path, tagged: "_ffi.DeviceReads", front: int, back: int, min_len: int) -> int:
    """``<sample>_umiCounts.csv`` (digest.py:183-197): one line ``UMI,insert,count`` per distinct UMI-tagged read whose insert
    has ``min_len`` bases, in the order the tagged reads first appeared; appended to, as the reference opens it.  Returns
    the number of lines.  Formatted with numpy from the flat arrays: no per-read Python object."""
    counts, _ = tagged.counts()
    full = tagged.unpack().take(tagged.first_appearance_order())
    c = counts[tagged.first_appearance_order(), 0] if len(tagged) else np.zeros(0, np.uint32)
    pure, tag = full.umi_split(front, back)
    keep = np.flatnonzero(pure.lengths >= int(min_len))
    pure, tag, c = pure.take(keep), tag.take(keep), c[keep]
    digits = FlatSeqs.from_fixed(np.char.mod("%d", c.astype(np.int64)).astype("S")) if len(c) else FlatSeqs.from_list([])
    with open(path, "ab") as fh:
        fh.write(b"UMISeq,transcriptSeq,UMICounts\n")
        fh.write(FlatSeqs.join_columns([tag, pure, digits], b",,\n"))
    return int(keep.shape[0])


def write_tcf(path, raw: "_ffi.DeviceReads") -> None:
    """``<sample>.trim.collapse.fa`` (``-tcf``, digest.py:219-229): the sample's own dictionary, most frequent first (Python's
    stable ``sorted(..., key=count, reverse=True)``: ties stay in dictionary order), ``>seq<k>_<count>``.  One collapse of the
    sample's raw reads on the GPU, the text assembled with numpy."""
    u1 = raw.collapse()
    cnt, first = u1.counts()
    c = cnt[:, 0].astype(np.int64) if len(u1) else np.zeros(0, np.int64)
    order = np.lexsort((first, -c))
    seqs = u1.unpack().take(order)
    u1.close()
    n = len(order)
    head = FlatSeqs.from_fixed(np.char.add(b">seq", np.char.mod("%d", np.arange(1, n + 1)).astype("S")).astype("S")) if n else FlatSeqs.from_list([])
    num = FlatSeqs.from_fixed(np.char.mod("%d", c[order]).astype("S")) if n else FlatSeqs.from_list([])
    with open(path, "wb") as fo:
        fo.write(FlatSeqs.join_columns([head, num, seqs], b"_\n\n"))


def parse_sample(ctx: _ffi.Context, text, min_len: int, trim, umi, workDir=None, name=None, timings: Dict[str, float] = None):
    """One file's text -> (raw reads as the collapse takes them, records seen): parse, the modifier chain, the length
    filter and -- with ``umi`` -- the reference's UMI handling, all on the GPU (``mirge_reads_parse[_trim|_umi]``).
    With ``-udd`` also writes ``<name>_umiCounts.csv``.  ``text`` may be a ``GzipRecordStream``: its pieces are uploaded and
    parsed one by one while the next ones inflate, and appended on the device (``mirge_reads_concat``: file order kept)."""
    if isinstance(text, (GzipRecordStream, TextRecordStream, ParallelGzipStream)):
        if umi is not None:  # the UMI routes look at the whole sample (a collapse inside): not streamed
            text = text.whole_text()
        else:
            return _parse_stream(ctx, text, min_len, trim, timings)
    if umi is None:
        return _ffi.DeviceReads.parse(ctx, text, 0, min_len, trim)
    raw, n_rec, tagged = _ffi.DeviceReads.parse_umi(ctx, text, 0, min_len, trim, umi)
    if tagged is not None:
        if workDir is not None:
            write_umi_counts(Path(workDir) / (str(name) + "_umiCounts.csv"), tagged, umi.front, umi.back, min_len)
        tagged.close()
    return raw, n_rec


def _parse_stream(ctx: _ffi.Context, stream, min_len: int, trim, timings=None):
    try:
        return _parse_stream_once(ctx, stream, min_len, trim, timings)
    except GzRouteDeclined:  # not a file for the parallel inflater: everything again through zlib (what was parsed ahead is closed)
        return _parse_stream_once(ctx, GzipRecordStream(stream.path), min_len, trim, timings)


def _parse_stream_once(ctx: _ffi.Context, stream, min_len: int, trim, timings=None):
    parts, n_rec = [], 0
    t_wait = t_gpu = 0.0
    t = time.perf_counter()
    try:
        for item in stream:
            t1 = time.perf_counter()
            t_wait += t1 - t
            if isinstance(item, tuple):  # not FASTQ: the whole text at once
                data = unwrap_fasta(item[1]) if item[1][:1] == b">" else item[1]
                r, n = _ffi.DeviceReads.parse(ctx, data, 0, min_len, trim)
            else:
                r, n = _ffi.DeviceReads.parse(ctx, item, 1, min_len, trim)
            parts.append(r)
            n_rec += n
            t = time.perf_counter()
            t_gpu += t - t1
        if not parts:
            raw = _ffi.DeviceReads.parse(ctx, b"", 0, min_len, trim)[0]
        elif len(parts) == 1:
            raw = parts.pop()
        else:
            raw = _ffi.DeviceReads.concat(ctx, parts)
    finally:
        stream.close()
        for r in parts:
            r.close()
    if timings is not None:
        timings["inflate_s"] = timings.get("inflate_s", 0.0) + stream.inflate_s
        timings["inflate_wait_s"] = timings.get("inflate_wait_s", 0.0) + t_wait   # the GPU side idle, waiting for text
        timings["upload_parse_s"] = timings.get("upload_parse_s", 0.0) + t_gpu
        timings["gz_text_MB"] = round(timings.get("gz_text_MB", 0.0) + stream.text_bytes / 1e6, 1)
        timings["gz_pieces"] = timings.get("gz_pieces", 0) + stream.pieces
    return raw, n_rec


def baking(args, inFileArray, inFileBaseArray, workDir, ctx: _ffi.Context = None):
    """Drop-in for ``baking`` (digest.py:105-302): every file's text goes to the GPU as it is -- records found, trimmed,
    UMI-sliced, filtered by length and packed there (digest.py:320-375 without a per-read host step); the samples' read
    sets are appended on the device and collapsed together with a sample id (digest.py:141-163 + the outer join :243)."""
    import pandas as pd
    trim = trim_from_args(args)
    umi = umi_from_args(args)
    begningTime = time.perf_counter()
    runlogFile = Path(workDir) / "run.log"
    outlog = open(str(runlogFile), "a+")
    ctx = ctx or _ffi.Context(getattr(args, "device", 0))
    min_len = int(getattr(args, "minimum_length", 16))
    sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique = {}, {}, {}
    parsed: List[_ffi.DeviceReads] = []
    texts = read_texts(inFileArray, stream=True)  # read ahead on worker threads; a .gz inflates piece by piece beside its parse
    for FQfile, name in zip(inFileArray, inFileBaseArray):
        start = time.perf_counter()
        raw, n_rec = parse_sample(ctx, next(texts), min_len, trim, umi, workDir, name)
        sampleReadCounts[name] = n_rec
        trimmedReadCounts[name] = len(raw)
        parsed.append(raw)
        finish2 = time.perf_counter()
        if not args.quiet:
            print(f'Cutadapt finished for file {name} in {round(finish2-start, 4)} second(s)')
        outlog.write(f'Cutadapt finished for file {name} in {round(finish2-start, 4)} second(s)\n')
        if getattr(args, "tcf_out", False):  # (:219-229): by count, ties in dict order
            write_tcf(Path(workDir) / (str(name) + '.trim.collapse.fa'), raw)
    t0 = time.perf_counter()
    if len(parsed) == 1:
        uniq = parsed[0].collapse()
    else:
        uniq = collapse_parsed_samples(ctx, parsed)
    for p in parsed:
        p.close()
    counts, first = uniq.counts()
    seqs = uniq.unpack().to_list()
    uniq.close()
    counts = counts.astype(np.int64)
    for s, name in enumerate(inFileBaseArray):
        trimmedReadCountsUnique[name] = int((counts[:, s] > 0).sum())
    if len(inFileBaseArray) == 1:
        order = np.argsort(first, kind="stable")  # dict insertion order
    else:
        order = np.argsort(np.asarray(seqs, dtype=object), kind="stable")  # outer join sorts the index
    complete_set = pd.DataFrame(counts[order], columns=list(inFileBaseArray),
                                index=pd.Index([seqs[i] for i in order], name='Sequence'))
    t1 = time.perf_counter()
    if not args.quiet:
        print(f'Collapsing finished in {round(t1-t0, 4)} second(s)\n')
    outlog.write(f'Collapsing finished in {round(t1-t0, 4)} second(s)\n')
    complete_set = complete_set.assign(**dict.fromkeys(PASS_COLUMNS, ''))
    complete_set = complete_set.assign(**dict.fromkeys(['annotFlag'], '0'))
    complete_set = complete_set.reindex(columns=['annotFlag'] + PASS_COLUMNS + list(inFileBaseArray))
    complete_set = complete_set.astype({"annotFlag": int})
    EndTime = time.perf_counter()
    if not args.quiet:
        print(f'Data pre-processing completed in {round(EndTime-begningTime, 4)} second(s)\n')
    outlog.write(f'\nData pre-processing completed in {round(EndTime-begningTime, 4)} second(s)\n\n')
    outlog.close()
    return (complete_set, sampleReadCounts, trimmedReadCounts, trimmedReadCountsUnique)
