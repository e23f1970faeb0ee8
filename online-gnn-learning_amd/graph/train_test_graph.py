"""Seed selection state of the stream: which labelled vertices are train / test, their replay priorities, and the
draws the three online strategies make from them.  Host-side logic only (no kernels).

Behavioural parity with R/train/graph/train_test_graph.py:12-248 — the graph-state surface the hot path uses
(SURVEY.md §8b): ``get_graph``, the two id maps, ``get_train_set``/``get_test_set``, ``get_new_train_nodes``,
``draw_random_train_nodes``, ``draw_priority_train_nodes``, ``update_priorities``, ``evolve``.  Kept quirks:
  * every snapshot's labelled arrivals are split 85/15 by ``sklearn.model_selection.train_test_split`` (numpy's
    global RNG), fewer than three arrivals all go to train;
  * arrivals enter the replay buffer at ``lo + 0.95 (hi - lo)`` of the priorities seen so far (``start_priority``
    while the buffer has seen nothing);
  * ``draw_priority_train_nodes(n)`` is a uniform shuffle prefix whenever ``n <= |train|`` — the buffer is only
    consulted for larger requests;
  * a priority update covering the whole train set rebuilds the buffer with the current (annealed) alpha.
Not carried over: the k-hop priority propagation to neighbours of new arrivals (``_get_affected_nodes`` +
``increment_priorities``) — in the reference that half of ``_update_priority_struct`` sits inside string literals
(R/train/graph/train_test_graph.py:98-135: the code between the triple quotes never runs), so the live behaviour is the
admission rule alone.

The buffer lives in HBM when the graph does (``prioritized_replay/device_buffer.py``): the PBR passes then
write their per-seed losses into it without a device->host transfer (``update_priorities_device``); ``device_replay=False``
keeps the host (numpy) buffer, which is also what a CPU-only graph object gets.
"""
from __future__ import annotations

import random

import numpy as np

from sklearn.model_selection import train_test_split

from ..prioritized_replay.replay_buffer import PrioritizedReplayBuffer
from ..prioritized_replay.device_buffer import DevicePrioritizedReplayBuffer

SIZE_BUFFER = 10000000   # nominal capacity the reference asks for; the sum tree here grows on demand


def _labelled(vertices, flags):
    return [v for v, is_labelled in zip(vertices, flags) if is_labelled]


class TrainTestGraph:
    def __init__(self, graph, split=0.25, start_prior_alpha=1, end_prior_alpha=2, scale=1, max_priority=3.0,
                 start_priority=2, min_priority=0.0000001, device_replay=None):
        self.temporal_graph = graph
        self.graph = graph.get_graph()
        dev = getattr(self.graph, "device", None)
        on_gpu = dev is not None and getattr(dev, "type", None) == "cuda"
        self.device_replay = on_gpu if device_replay is None else (bool(device_replay) and on_gpu)
        self.size_evolution = len(graph)
        self.split, self.scale = split, scale
        self.start_prior_alpha, self.end_prior_alpha = start_prior_alpha, end_prior_alpha
        self.prior_alpha = start_prior_alpha
        self.max_priority, self.min_priority, self.start_priority = max_priority, min_priority, start_priority
        self.train_set, self.test_set = set(), set()
        self._train_list = self._test_list = None        # list(set), rebuilt on first use after an admission (see the properties)
        self._train_arr, self._train_n = np.empty(1024, dtype=np.int64), 0     # the train set in ARRIVAL order (get_train_array)
        self.train, self.test = [], []                   # the arrivals of the latest snapshot only
        self.exact_shuffle = False                       # see _shuffle_prefix
        self.priority_replay_buffer = self._fresh_buffer()
        self._admit(_labelled(*graph.get_added_vertices()))

    # ---- internal -------------------------------------------------------------------------------------------
    def _fresh_buffer(self):
        if self.device_replay:
            return DevicePrioritizedReplayBuffer(SIZE_BUFFER, self.prior_alpha, max_priority=self.max_priority,
                                                 min_priority=self.min_priority, device=self.graph.device,
                                                 key_space=getattr(self.graph, "n_total", 1024))
        return PrioritizedReplayBuffer(SIZE_BUFFER, self.prior_alpha, max_priority=self.max_priority,
                                       min_priority=self.min_priority)

    def _admit(self, arrivals):
        """Split one snapshot's labelled arrivals and enrol the train part in the replay buffer."""
        if len(arrivals) < 3:
            self.train, self.test = set(arrivals), set()
        else:
            self.train, self.test = train_test_split(arrivals, shuffle=True, test_size=self.split)
        fresh = [v for v in dict.fromkeys(self.train) if v not in self.train_set]
        if self._train_n + len(fresh) > self._train_arr.size:
            grown = np.empty(max(2 * self._train_arr.size, self._train_n + len(fresh)), dtype=np.int64)
            grown[:self._train_n] = self._train_arr[:self._train_n]
            self._train_arr = grown
        self._train_arr[self._train_n:self._train_n + len(fresh)] = fresh
        self._train_n += len(fresh)
        self.train_set |= set(self.train)
        self.test_set |= set(self.test)
        self._train_list = self._test_list = None
        self._update_priority_struct()

    # ``list(self.train_set)`` / ``list(self.test_set)`` as the reference keeps them (R/train/graph/train_test_graph.py:78-96: rebuilt on
    # every admission) — here on first use after an admission: a snapshot that never asks for the list (a PBR snapshot whose priority
    # forward walks ``get_train_array()``) does not pay the O(|train set|) rebuild, 1.2 ms at the arxiv-like stream's 136 k vertices.
    @property
    def train_set_list(self):
        if self._train_list is None:
            self._train_list = list(self.train_set)
        return self._train_list

    @property
    def test_set_list(self):
        if self._test_list is None:
            self._test_list = list(self.test_set)
        return self._test_list

    def get_train_array(self):
        """The train set as an int64 array of original vertex ids in ARRIVAL order, maintained incrementally (O(arrivals) per snapshot).
        For whole-set passes that only need every member once — the PBR priority forward: converting the 136 k-entry Python list of the
        arxiv-like stream to an array and back cost 9 ms of a 17 ms snapshot.  (The list's own order is CPython's set order; a pass
        over the whole set is indifferent to it.)"""
        out = self._train_arr[:self._train_n]
        out.flags.writeable = False
        return out

    def _update_priority_struct(self):
        buf = self.priority_replay_buffer
        fresh = list(dict.fromkeys(self.train))
        if self.device_replay:                           # the admission priority is derived from the extrema on the device
            buf.admit(np.asarray(fresh, dtype=np.int64), self.start_priority) if fresh else None
            return
        hi = buf.get_max_priority()
        if hi == -1:                                     # nothing scored yet
            entry = self.start_priority
        else:
            lo = buf.get_min_priority()
            entry = lo + (hi - lo) * 0.95
        buf.add_all_arrays(np.asarray(fresh, dtype=np.int64), np.full(len(fresh), float(entry))) if fresh else None

    # ---- stream ---------------------------------------------------------------------------------------------
    def __len__(self):
        return len(self.temporal_graph)

    def evolve(self):
        span = self.end_prior_alpha - self.start_prior_alpha
        self.prior_alpha = self.start_prior_alpha + (span / len(self)) * self.temporal_graph.evolution_index
        self.temporal_graph.evolve()
        self.graph = self.temporal_graph.get_graph()
        self._admit(_labelled(*self.temporal_graph.get_added_vertices()))

    def get_graph(self):
        return self.temporal_graph.get_graph()

    def get_original_to_subgraph_map(self):
        return self.temporal_graph.get_original_to_subgraph_map()

    def get_subgraph_to_original_map(self):
        return self.temporal_graph.get_subgraph_to_original_map()

    # ---- sets and draws -------------------------------------------------------------------------------------
    def get_train_set(self):
        return self.train_set_list

    def get_test_set(self):
        return self.test_set_list

    def get_new_test_nodes(self):
        return self.test

    def get_new_train_nodes(self, batch_size=None):
        fresh = list(self.train)
        if batch_size is not None and batch_size < len(fresh):
            random.shuffle(fresh)
            del fresh[batch_size:]
        return fresh

    def _shuffle_prefix(self, n_nodes):
        """Uniform random ``n_nodes``-subset of the train set in random order.  The reference shuffles the WHOLE list
        and slices (``random.shuffle`` of 2e5 ids, 100 times per Reddit snapshot: ~0.1 s each); a partial Fisher-Yates
        pass over the first ``n_nodes`` positions yields the same distribution in O(n_nodes).  ``exact_shuffle=True``
        restores the reference's RNG consumption (identical draws under the same ``random.seed``)."""
        total = len(self.train_set)
        if not self.exact_shuffle and n_nodes * 8 < total:
            # a small subset of a long list (512 of 2e5, 100 times per Reddit snapshot): an ordered uniform sample of positions
            # from a numpy Generator (Floyd's algorithm: O(n_nodes), in C) seeded ONCE from Python's `random` stream — the stream
            # the reference draws from, so identically seeded replicas still draw identical batches.  The positions index the
            # arrival-order array (a uniform subset whatever the order): the list form — rebuilt after every admission, 0.7 ms at the
            # arxiv-like stream's 136 k vertices — is not touched (round 6: it was most of a PBR snapshot's host time outside the forward)
            if getattr(self, "_draw_rng", None) is None:
                self._draw_rng = np.random.default_rng(random.getrandbits(63))
            idx = self._draw_rng.choice(total, size=n_nodes, replace=False, shuffle=True)
            return self._train_arr[:self._train_n][idx].tolist()
        lst = self.train_set_list
        if self.exact_shuffle:
            random.shuffle(lst)
            return lst[:n_nodes]
        for i in range(n_nodes):
            j = random.randrange(i, total)
            lst[i], lst[j] = lst[j], lst[i]
        return lst[:n_nodes]

    def draw_random_train_nodes(self, n_nodes):
        return self._shuffle_prefix(n_nodes) if n_nodes <= len(self.train_set) else self.train_set_list

    def draw_priority_train_nodes(self, n_nodes):
        if n_nodes <= len(self.train_set):
            return self._shuffle_prefix(n_nodes)
        return self.priority_replay_buffer.sample(n_nodes)

    # ---- priorities -----------------------------------------------------------------------------------------
    def dump_priorities(self, vertex_list):
        return self.priority_replay_buffer.dump_priorities(vertex_list)

    def update_priorities_arrays(self, ids, priorities):
        """update_priorities(dict(zip(ids, priorities))) without the dict: distinct original vertex ids as an array."""
        ids = np.asarray(ids)
        assert len(ids) <= len(self.train_set)
        if len(ids) == len(self.train_set):
            self.priority_replay_buffer = self._fresh_buffer()
            self.priority_replay_buffer.add_all_arrays(ids, priorities)
        else:
            self.priority_replay_buffer.update_arrays(ids, priorities)

    def update_priorities_device(self, ids, priorities_dev):
        """``update_priorities_arrays`` with the priorities still on the device (a float32 / float64 CUDA tensor, e.g. the
        per-seed losses of a PBR pass) and the distinct original vertex ids on the host: nothing is copied back."""
        assert self.device_replay, "update_priorities_device needs the device-side buffer"
        ids = np.asarray(ids, dtype=np.int64)
        assert len(ids) <= len(self.train_set) and priorities_dev.numel() == len(ids)
        if len(ids) == len(self.train_set):
            self.priority_replay_buffer = self._fresh_buffer()
            self.priority_replay_buffer.add_all_arrays(ids, priorities_dev)
        else:
            import torch
            self.priority_replay_buffer.update_device(torch.as_tensor(ids).to(priorities_dev.device, non_blocking=True), priorities_dev)

    def update_priorities(self, d_priorities):
        """{original vertex id: new priority}.  A partial update rewrites leaves; a full one rebuilds the buffer."""
        assert len(d_priorities) <= len(self.train_set)
        if len(d_priorities) == len(self.train_set):
            self.priority_replay_buffer = self._fresh_buffer()
            self.priority_replay_buffer.add_all(d_priorities)
        else:
            self.priority_replay_buffer.update_priorities(d_priorities)
