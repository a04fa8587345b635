"""Background writer of the ``.npz`` artefacts of the samplers.

The reference writes ``imputations_{b}.npz`` / ``irwg_imputations_after_final_resampling_{b}.npz``
with ``np.savez_compressed`` at the end of every batch (test_step_vae_sampling.py:785-791,
2039-2045).  At the published settings one file holds 1-2 GB of float32 and its zlib compression
takes several times longer than the whole sampling loop of the batch on the GPU, so the files are
written by a small pool of host threads (zlib releases the GIL) while the next batch is sampled.
File names, keys and contents are unchanged.  At most ``max_pending`` files are in flight (their
arrays are held in host memory until written).
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class AsyncNpzWriter:
    def __init__(self, workers=4, max_pending=4):
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="npz-writer")
        self._slots = threading.Semaphore(max_pending)
        self._futures = []

    def submit(self, path, **arrays):
        """``np.savez_compressed(path, **arrays)`` on a worker thread; blocks while ``max_pending``
        earlier files are still being written."""
        self._slots.acquire()

        def job():
            try:
                np.savez_compressed(path, **arrays)
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
