"""Device-resident imputation histories: ``ImputationHistoryQueue`` (LAIR's historical proposals,
irwg/sampling/utils.py:5-68) and the restricted-availability queue of AC-MWG.

The latter is the counterpart of ``ImputationHistoryQueue_with_restrictedavailability``
(irwg/sampling/utils.py:83-237).  The reference keeps the ring ``[T,B,K,D]`` and the per-chain
availability counters on the host and draws one index per chain in a Python double loop; here
both live in HBM, the index draw / gather are fused into the AC-MWG step kernels
(``cs_acmwg_step``) and this class only tracks the two host scalars (``tail_idx``, ``full``)
and exposes the queue's methods for the warm-up phases and for tests.
"""
import ctypes as C

import torch

from .. import _lib


class ImputationHistoryQueue(object):
    """Ring ``[T,B,K,D]`` of past imputations in HBM with the reference's stratified sampling
    (irwg/sampling/utils.py:5-68).  The host tracks ``tail_idx`` / ``full`` and computes the H ring
    slots of a draw (they are common to the batch); the gather of the chosen imputations into the
    proposal set runs on the device (``cs_history_concat``)."""

    def __init__(self, max_history_length, batch_shape, dtype=torch.float32, device="cuda"):
        if dtype not in (torch.float32, torch.bool):
            raise NotImplementedError(f"history dtype {dtype}")
        self.max_history_length = int(max_history_length)
        self.tail_idx = -1
        self.full = False
        self.batch_shape = tuple(batch_shape)
        B, K, D = self.batch_shape
        # dtype bool (the reference's queue for binary data, which stores X.bool()): a uint8 ring holding x != 0
        self.u8 = dtype == torch.bool
        self.history_queue = torch.empty((self.max_history_length, B, K, D),
                                         dtype=torch.uint8 if self.u8 else torch.float32, device=device)

    def __len__(self):
        assert self.tail_idx > -1, 'Queue is empty.'
        return self.tail_idx + 1 if not self.full else self.max_history_length

    def _increment_tail(self):
        self.tail_idx = (self.tail_idx + 1) % self.max_history_length
        if self.tail_idx == self.max_history_length - 1:
            self.full = True

    def enqueue_batch(self, X):
        B, K, D = self.batch_shape
        X = X.to(self.history_queue.device, torch.float32).contiguous()
        self._increment_tail()
        fn = _lib.lib().cs_history_enqueue_u8 if self.u8 else _lib.lib().cs_history_enqueue
        _lib.check(fn(_lib.ptr(self.history_queue), None, self.max_history_length, self.tail_idx, 0, _lib.ptr(X), B, K, D,
                      _lib.current_stream()))

    def window_size(self, num_historical_proposals):
        """Exclusive bound of the per-window integer draws (utils.py:33-36)."""
        return len(self) // num_historical_proposals

    def slots_from_window_draws(self, r, num_historical_proposals):
        """Ring slots of the window draws r[H] (utils.py:37-39)."""
        H, w = num_historical_proposals, self.window_size(num_historical_proposals)
        return [int((int(r[h]) + h * w + self.tail_idx) % len(self)) for h in range(H)]

    def concat_history(self, X, slots, *, chain_of_proposal=None, chain_of_example=None, rng=None, out=None):
        """``concat([X, hist_proposals], dim=1)`` of vae_imputation.py:539-545 -> [B,K+H,D]: historical
        proposal h of example b is ring[slots[h], b, chain] with chain = chain_of_proposal[h]
        (sample_history_with_chain_idx_for_each_sample) or one chain per example
        (sample_history: ``chain_of_example`` [B] int64 on the device, or the counter RNG)."""
        B, K, D = self.batch_shape
        H = len(slots)
        Xc = out if out is not None else torch.empty(B, K + H, D, device=X.device, dtype=torch.float32)
        i32 = C.c_int32 * H
        cop = i32(*[int(c) for c in chain_of_proposal]) if chain_of_proposal is not None else None
        coe = chain_of_example.to(X.device, torch.int64).contiguous() if chain_of_example is not None else None
        rng = rng if rng is not None else _lib.make_rng(0)
        fn = _lib.lib().cs_history_concat_u8 if self.u8 else _lib.lib().cs_history_concat
        _lib.check(fn(_lib.ptr(X), _lib.ptr(self.history_queue), B, K, D, H, i32(*slots), cop, _lib.ptr(coe), C.byref(rng),
                      _lib.ptr(Xc), _lib.current_stream()))
        return Xc


def glr_history_chain_idx(K, num_historical_proposals, glr_cluster_size):
    """Chain that every historical proposal is taken from when global-local resampling is combined
    with the history, so that each group only receives history of its own chains
    (vae_imputation.py:505-537)."""
    import math
    H, c = num_historical_proposals, glr_cluster_size
    assert (K + H) % c == 0
    G = (K + H) // c
    assert G <= K
    num_rem = math.ceil(K / G) * G - K
    base = list(range(K)) + list(range(G - num_rem, G))
    chain = base * ((K + H) // len(base))
    if (K + H) % len(base) != 0:
        chain = chain + base[:(K + H) - len(chain)]
    return chain[K:]


class ImputationHistoryQueue_with_restrictedavailability(object):
    def __init__(self, max_history_length, batch_shape, dtype=torch.float32, device="cuda", compact=True):
        if dtype not in (torch.float32, torch.bool):
            raise NotImplementedError(f"history dtype {dtype}")
        self.max_history_length = int(max_history_length)
        self.tail_idx = -1
        self.full = False
        self.bool_data = dtype == torch.bool  # values are 0/1
        # dtype bool (the reference stores X.bool(), test_step_vae_sampling.py:1537-1541): a uint8 ring holding x != 0,
        # a quarter of the float ring (Omniglot, B=200, K=5, T=5000: 3.9 GB instead of 15.7).  compact=False keeps
        # the float ring: the fused MLP encoder reads its input rows as float32 straight from the ring.
        self.u8 = self.bool_data and compact
        self.batch_shape = tuple(batch_shape)
        B, K, D = self.batch_shape
        # t b k d
        self.history_queue = torch.empty((self.max_history_length, B, K, D),
                                         dtype=torch.uint8 if self.u8 else torch.float32, device=device)
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
        fn = _lib.lib().cs_history_enqueue_u8 if self.u8 else _lib.lib().cs_history_enqueue
        _lib.check(fn(_lib.ptr(self.history_queue), _lib.ptr(self.available_timesteps), self.max_history_length, new_tail,
                      int(was_full), _lib.ptr(X), B, K, D, _lib.current_stream()))

    def slots_from_draws(self, r):
        """Ring slot of integer draws r[B,K] in [0, max(avail,1)) (utils.py:225-232)."""
        if self.full:
            return (r + self.tail_idx + 1) % self.max_history_length
        return r
