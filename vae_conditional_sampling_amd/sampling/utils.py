"""Device-resident imputation history for AC-MWG.

Counterpart of ``ImputationHistoryQueue_with_restrictedavailability``
(irwg/sampling/utils.py:83-237).  The reference keeps the ring ``[T,B,K,D]`` and the per-chain
availability counters on the host and draws one index per chain in a Python double loop; here
both live in HBM, the index draw / gather are fused into the AC-MWG step kernels
(``cs_acmwg_step``) and this class only tracks the two host scalars (``tail_idx``, ``full``)
and exposes the queue's methods for the warm-up phases and for tests.
"""
import ctypes as C

import torch

from .. import _lib


class ImputationHistoryQueue_with_restrictedavailability(object):
    def __init__(self, max_history_length, batch_shape, dtype=torch.float32, device="cuda"):
        if dtype not in (torch.float32, torch.bool):
            raise NotImplementedError(f"history dtype {dtype}")
        self.max_history_length = int(max_history_length)
        self.tail_idx = -1
        self.full = False
        self.bool_data = dtype == torch.bool  # values are 0/1; the ring itself is float32
        self.batch_shape = tuple(batch_shape)
        B, K, D = self.batch_shape
        # t b k d
        self.history_queue = torch.empty((self.max_history_length, B, K, D), dtype=torch.float32, device=device)
        # b k
        self.available_timesteps = torch.zeros((B, K), dtype=torch.int64, device=device)

    def __len__(self):
        assert self.tail_idx > -1, 'Queue is empty.'
        return self.tail_idx + 1 if not self.full else self.max_history_length

    def update_available_timesteps(self, make_full_history_available):
        """BxK boolean tensor (or None = all chains): those chains may use the whole history."""
        mask = make_full_history_available
        m8 = None
        if mask is not None:
            m8 = mask.to(self.available_timesteps.device, torch.uint8).contiguous()
        B, K, _ = self.batch_shape
        _lib.check(_lib.lib().cs_history_set_available(_lib.ptr(self.available_timesteps), _lib.ptr(m8), B * K,
                                                       len(self), _lib.current_stream()))

    def advance(self):
        """Host bookkeeping of one enqueue (utils.py:109-113); returns (new_tail, was_full)."""
        was_full = self.full
        self.tail_idx = (self.tail_idx + 1) % self.max_history_length
        if self.tail_idx == self.max_history_length - 1:
            self.full = True
        return self.tail_idx, was_full

    def enqueue_batch(self, X):
        B, K, D = self.batch_shape
        X = X.to(self.history_queue.device, torch.float32).contiguous()
        new_tail, was_full = self.advance()
        _lib.check(_lib.lib().cs_history_enqueue(_lib.ptr(self.history_queue), _lib.ptr(self.available_timesteps),
                                                 self.max_history_length, new_tail, int(was_full), _lib.ptr(X),
                                                 B, K, D, _lib.current_stream()))

    def slots_from_draws(self, r):
        """Ring slot of integer draws r[B,K] in [0, max(avail,1)) (utils.py:225-232)."""
        if self.full:
            return (r + self.tail_idx + 1) % self.max_history_length
        return r
