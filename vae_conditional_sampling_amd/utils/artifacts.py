"""Background writer of the ``.npz`` artefacts of the samplers.

The reference writes ``imputations_{b}.npz`` / ``irwg_imputations_after_final_resampling_{b}.npz``
with ``np.savez_compressed`` at the end of every batch (test_step_vae_sampling.py:785-791,
2039-2045).  At the published settings one file holds 1-2 GB of float32 and its zlib compression
takes several times longer than the whole sampling loop of the batch on the GPU, so the files are
written by a small pool of host threads (zlib releases the GIL) while the next batch is sampled.
File names, keys and contents are unchanged.  At most ``max_pending`` files are in flight (their
arrays are held in host memory until written).

One file is mostly one array (``imputations``), so a thread per file still leaves each file to a single core
for a minute: ``savez_compressed_parallel`` writes the same zip container (deflate members, readable by
``np.load``) with every member's deflate stream produced in independent pieces on a pool of threads, the way
parallel gzip implementations do (each piece a raw deflate run closed by a full flush, an empty final block at
the end: the concatenation is one valid stream).

Large float arrays whose rows are short (the kept imputations [T,B,K,D]) go through the library's own deflate
encoder instead of zlib (``cs_deflate_rows``: matches one row back only, a Huffman code per block): the same
container and the same file sizes to within a percent, at several times zlib's speed per thread, which is what the
end-to-end time of the published configurations was bound by.
"""
import ctypes
import os
import threading
import zipfile
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .. import _lib

_PIECE = 4 << 20          # bytes compressed per task
_SLICE = 64 * _PIECE      # bytes handed to one write() call
_ROW_ENCODER_MIN_BYTES = 1 << 20


def deflate_rows(data, stride, lookback):
    """One piece of a raw deflate stream for the bytes of ``data`` (a contiguous buffer), matches at distance
    ``stride`` only (cs_deflate_rows); ``lookback`` bytes before it in memory belong to the same stream."""
    src = np.frombuffer(data, dtype=np.uint8)
    n = src.size
    dst = np.empty(n + n // 8 + 64, dtype=np.uint8)
    r = _lib.lib().cs_deflate_rows(src.ctypes.data, n, int(stride), int(lookback), dst.ctypes.data, dst.size)
    if r < 0:
        raise _lib.CsError(f"cs_deflate_rows failed: {_lib.lib().cs_last_error().decode()}")
    return dst[:r].tobytes()


def row_stride_of(arr):
    """Row length in bytes when ``arr`` is an array the row encoder is meant for, else None."""
    if arr.dtype.kind != "f" or arr.ndim < 2 or arr.nbytes < _ROW_ENCODER_MIN_BYTES:
        return None
    stride = arr.shape[-1] * arr.itemsize
    return stride if 1 <= stride <= 32768 else None


def _crc32(data):
    """CRC-32 of a contiguous buffer (cs_crc32: the zip container's checksum, without the interpreter lock)."""
    src = np.frombuffer(data, dtype=np.uint8)
    return _lib.lib().cs_crc32(src.ctypes.data if src.size else None, src.size, 0)


class _PieceDeflate:
    """Stands in for the compressor object of an open zip member: ``compress`` deflates its input in
    independent ``_PIECE``-sized pieces on the pool, ``flush`` closes the stream.  The threads also checksum their
    pieces; ``crc`` is the checksum of everything compressed so far, continued from ``crc0``."""

    def __init__(self, pool, level, crc0=0):
        self._pool, self._level, self.crc, self._done = pool, level, crc0, 0

    def _deflate(self, piece, offset):
        c = zlib.compressobj(self._level, zlib.DEFLATED, -15)
        return c.compress(piece) + c.flush(zlib.Z_FULL_FLUSH)

    def _one(self, args):
        piece, offset = args
        return self._deflate(piece, offset), _crc32(piece), len(piece)

    def compress(self, data, offset=None):
        mv = memoryview(data)
        offset = self._done if offset is None else offset   # (a caller that only knows write(): consecutive slices)
        self._done = offset + len(mv)
        parts = list(self._pool.map(self._one, [(mv[i:i + _PIECE], offset + i) for i in range(0, len(mv), _PIECE)]))
        for _, crc, n in parts:
            self.crc = _lib.lib().cs_crc32_combine(self.crc, crc, n)
        return b"".join(p[0] for p in parts)

    def flush(self):
        return zlib.compressobj(self._level, zlib.DEFLATED, -15).flush(zlib.Z_FINISH)


class _RowDeflate(_PieceDeflate):
    """The same for an array of short rows: the pieces go through cs_deflate_rows.  ``compress`` is called with
    consecutive slices of one contiguous buffer (``offset``: where the slice starts in it), so a piece may match into
    the bytes before it."""

    def __init__(self, pool, level, stride, crc0=0):
        super().__init__(pool, level, crc0)
        self._stride = stride

    def _deflate(self, piece, offset):
        return deflate_rows(piece, self._stride, offset)


def savez_compressed_parallel(path, pool, level=6, **arrays):
    """``np.savez_compressed(path, **arrays)`` with the compression of every member spread over ``pool``."""
    path = os.fspath(path)
    if not path.endswith(".npz"):
        path += ".npz"
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED, allowZip64=True) as zf:
        for key, val in arrays.items():
            arr = np.asanyarray(val)
            with zf.open(key + ".npy", "w", force_zip64=True) as f:
                if arr.dtype.hasobject or getattr(f, "_compressor", None) is None:
                    np.lib.format.write_array(f, arr, allow_pickle=True)     # numpy's own path
                    continue
                arr = arr if arr.flags.c_contiguous else np.ascontiguousarray(arr)   # (keeps 0-d arrays 0-d)
                f._compressor = _PieceDeflate(pool, level)
                np.lib.format.write_array_header_1_0(f, np.lib.format.header_data_from_array_1_0(arr))
                # From here on the stream is the array's memory.  The member's bookkeeping (what ZipFile's write()
                # does, with the checksum taken by one thread over the whole array) is done here from the pieces'.
                stride = row_stride_of(arr)
                crc0 = getattr(f, "_crc", 0)
                comp = _RowDeflate(pool, level, stride, crc0) if stride else _PieceDeflate(pool, level, crc0)
                f._compressor = comp
                raw = memoryview(arr.reshape(-1)).cast("B") if arr.size else memoryview(b"")
                own_books = all(hasattr(f, a) for a in ("_crc", "_file_size", "_compress_size", "_fileobj"))
                for i in range(0, len(raw), _SLICE):
                    if not own_books:        # another zipfile implementation: its write() keeps the books
                        f.write(raw[i:i + _SLICE])
                        continue
                    out = comp.compress(raw[i:i + _SLICE], i)
                    f._file_size += len(raw[i:i + _SLICE])
                    f._compress_size += len(out)
                    f._fileobj.write(out)
                if own_books:
                    f._crc = comp.crc


class AsyncNpzWriter:
    def __init__(self, workers=4, max_pending=4, compress_threads=None):
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="npz-writer")
        # threads that deflate the pieces of the files being written (zlib releases the GIL); OMP_NUM_THREADS is
        # the CPU share of a process on a shared host
        n = compress_threads or max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", 0)) or (os.cpu_count() or 1)))
        self._deflate_pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="npz-deflate")
        self._slots = threading.Semaphore(max_pending)
        self._futures = []

    def submit(self, path, **arrays):
        """``np.savez_compressed(path, **arrays)`` (its parallel form) on a worker thread; blocks while
        ``max_pending`` earlier files are still being written."""
        self._slots.acquire()

        def job():
            try:
                savez_compressed_parallel(path, self._deflate_pool, **arrays)
            finally:
                self._slots.release()
        self._futures.append(self._pool.submit(job))

    def wait(self):
        """Block until every submitted file is on disk; re-raises the first writer error."""
        futures, self._futures = self._futures, []
        for f in futures:
            f.result()

    def close(self):
        self.wait()
        self._pool.shutdown(wait=True)
        self._deflate_pool.shutdown(wait=True)
