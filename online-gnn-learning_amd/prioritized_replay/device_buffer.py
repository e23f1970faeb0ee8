"""Prioritised replay buffer in HBM: the device-side twin of ``replay_buffer.PrioritizedReplayBuffer``.

Same behaviour as R/train/prioritized_replay/replay_buffer.py:59-283 + segment_tree.py:69-125 (pinned by
tests/golden/replay.json through the host class and, for this one, by tests/test_gpu_replay.py): clip to
[min_priority, max_priority], log, min-max normalise with RUNNING extrema, offset (+1e-5 insert / +1e-6 update), ``** alpha``;
stratified-proportional sampling with de-duplication, up to 21 proportional re-draws, uniform top-up; the end-exclusive
``p_total`` quirk.  What lives where:

* HBM: the fp64 sum tree (``node``), the running extrema (``state``), the vertex-id -> leaf map (``idx_map``), the stored ids.
  The per-seed losses of the PBR train update and of the priority forward are written into the tree by ONE kernel
  (``ogl_replay_update``) from the device tensors they were computed into: no ``.cpu()``, no Python loop, no host tree.
* host: Python's ``random`` stream (the reference draws its uniforms from it, so the uniforms are generated there and the
  tree walks of a whole batch run in one kernel, ``ogl_replay_sample``) and the O(batch) set logic of
  ``_sample_proportional``; the stored ids as a list (``sample`` returns them).  Sampling from the buffer is the cold path:
  ``TrainTestGraph.draw_priority_train_nodes`` consults it only when a request exceeds the train set.

fp64 ``log`` / ``pow`` on the device are not bit-identical to glibc's, so leaves agree with the host class to ~1e-15
relative, not bit for bit; sampled index sets agree unless a mass lands within that of a leaf boundary.
"""
from __future__ import annotations

import ctypes as C
import random

import numpy as np
import torch

from .. import _lib
from ..ops import _ptr, _stream
from .._lib import check


class DevicePrioritizedReplayBuffer:
    def __init__(self, size, alpha, max_priority, min_priority, device="cuda", key_space=1024):
        assert alpha >= 0
        self._maxsize = size
        self._alpha = alpha
        self._max_clip_priority, self._min_clip_priority = max_priority, min_priority
        self.device = torch.device(device)
        self._storage = []                                   # host list of keys, insertion order
        self._next_idx = 0
        self.cap = 1024
        self.node = torch.zeros(2 * self.cap, dtype=torch.float64, device=self.device)
        self.state = torch.tensor([-1.0, 99999999.0, -1.0, 99999999.0], dtype=torch.float64, device=self.device)
        self.idx_map = torch.full((max(int(key_space), 1),), -1, dtype=torch.int64, device=self.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._dict = {}
        self._dict_upto = 0

    def __len__(self):
        return len(self._storage)

    # ---- internals -----------------------------------------------------------------------------------------------
    @property
    def _key_to_idx(self):
        if self._dict_upto < len(self._storage):
            d, st = self._dict, self._storage
            for i in range(self._dict_upto, len(st)):
                d[st[i]] = i
            self._dict_upto = len(st)
        return self._dict

    def _grow(self, need):
        cap = self.cap
        while cap < need:
            cap *= 2
        if cap == self.cap:
            return
        node = torch.zeros(2 * cap, dtype=torch.float64, device=self.device)
        node[cap:cap + self.cap] = self.node[self.cap:2 * self.cap]
        self.node, self.cap = node, cap
        check(_lib.lib().ogl_replay_rebuild(_ptr(self.node), self.cap, _stream()), "ogl_replay_rebuild")

    def _grow_map(self, need):
        if need > self.idx_map.numel():
            grown = torch.full((max(need, 2 * self.idx_map.numel()),), -1, dtype=torch.int64, device=self.device)
            grown[:self.idx_map.numel()] = self.idx_map
            self.idx_map = grown

    def _update(self, idx_dev, prio_dev, offset, start_priority=0.0):
        n = int(idx_dev.numel())
        if n == 0:
            return
        p32 = prio_dev if (prio_dev is not None and prio_dev.dtype == torch.float32) else None
        p64 = prio_dev if (prio_dev is not None and prio_dev.dtype == torch.float64) else None
        if prio_dev is not None:
            assert (p32 is not None or p64 is not None) and prio_dev.is_cuda and prio_dev.is_contiguous() and prio_dev.numel() == n
        scratch = torch.empty(n, dtype=torch.float64, device=self.device)
        check(_lib.lib().ogl_replay_update(_ptr(self.node), self.cap, _ptr(idx_dev), _ptr(p32), _ptr(p64), n,
                                           C.c_double(float(self._min_clip_priority)), C.c_double(float(self._max_clip_priority)),
                                           C.c_double(offset), C.c_double(float(self._alpha)), C.c_double(float(start_priority)),
                                           _ptr(self.state), _ptr(scratch), _ptr(self.err), _stream()), "ogl_replay_update")

    def check_errors(self):
        """The reference asserts ``v >= 0`` per element; here the kernel raises a flag that is read when asked for."""
        e = int(self.err.item())
        assert e == 0, "replay update: negative / NaN scaled priority (1) or leaf index out of range (2): %d" % e

    # ---- insertion / update ---------------------------------------------------------------------------------------
    def _append_keys(self, keys_host):
        keys_host = np.ascontiguousarray(keys_host, dtype=np.int64)
        start = self._next_idx
        n = int(keys_host.size)
        self._storage.extend(keys_host.tolist())
        self._next_idx += n
        self._grow(self._next_idx)
        kd = torch.as_tensor(keys_host).to(self.device)
        if n and int(keys_host.min()) >= 0:
            self._grow_map(int(keys_host.max()) + 1)
            check(_lib.lib().ogl_replay_note_keys(_ptr(kd), n, start, _ptr(self.idx_map), self.idx_map.numel(), _stream()),
                  "ogl_replay_note_keys")
        return torch.arange(start, start + n, dtype=torch.int64, device=self.device)

    def add_all_arrays(self, keys, priorities):
        """add_all({keys[i]: priorities[i]}) for distinct non-negative integer keys; ``priorities`` host array or device tensor."""
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        if keys.size == 0:
            return
        idx = self._append_keys(keys)
        pr = priorities if isinstance(priorities, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(priorities, dtype=np.float64))
        self._update(idx, pr.to(self.device).contiguous(), 0.00001)

    def add_all(self, node_priority_dict):
        if node_priority_dict:
            self.add_all_arrays(list(node_priority_dict.keys()), [node_priority_dict[k] for k in node_priority_dict])

    def admit(self, keys, start_priority):
        """Enrol new train vertices at the admission priority (train_test_graph.py:78-93) — computed ON THE DEVICE from the
        running extrema, so the snapshot's admission needs no read-back of them."""
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        if keys.size == 0:
            return
        self._update(self._append_keys(keys), None, 0.00001, start_priority=start_priority)

    def update_device(self, keys_dev, priorities_dev):
        """update_priorities({keys: priorities}) with both on the device: the PBR passes' per-seed losses go straight in."""
        keys_dev = keys_dev.reshape(-1)
        if keys_dev.numel() == 0:
            return
        assert keys_dev.dtype == torch.int64 and keys_dev.is_cuda
        from .. import ops
        idx = ops.gather_i64(self.idx_map, keys_dev.contiguous())           # -1 for a key that is not in the buffer -> err flag 2
        self._update(idx, priorities_dev.reshape(-1).contiguous(), 0.000001)

    def update_arrays(self, keys, priorities):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        if keys.size == 0:
            return
        self.update_device(torch.as_tensor(keys).to(self.device),
                           torch.as_tensor(np.ascontiguousarray(priorities, dtype=np.float64)).to(self.device))

    def update_priorities(self, d_priorities):
        if d_priorities:
            self.update_arrays(list(d_priorities.keys()), [d_priorities[k] for k in d_priorities])

    # ---- sampling ---------------------------------------------------------------------------------------------------
    def _sample_proportional(self, batch_size):
        n = len(self._storage)
        if batch_size >= n:
            return list(self._key_to_idx.values())
        # the reference consumes `batch_size` uniforms, then one per re-draw while the set is short (at most 21), then
        # randint top-ups: generate the certain ones, SPECULATE the 21 re-draw uniforms, and rewind Python's stream to
        # what the reference would have consumed
        u = [random.random() for _ in range(batch_size)]
        saved = random.getstate()
        ur = [random.random() for _ in range(21)]
        random.setstate(saved)
        ud = torch.tensor(u + ur, dtype=torch.float64).to(self.device)
        out = torch.empty(batch_size + 21, dtype=torch.int64, device=self.device)
        ptot = torch.empty(1, dtype=torch.float64, device=self.device)
        check(_lib.lib().ogl_replay_sample(_ptr(self.node), self.cap, n, batch_size, _ptr(ud), _ptr(ud[batch_size:]), 21, _ptr(out),
                                           _ptr(ptot), _stream()), "ogl_replay_sample")
        picks = out.cpu().tolist()
        res = set()
        for i in range(batch_size):
            res.add(picks[i])
        tries = 0
        while len(res) < batch_size:
            random.random()                                  # the uniform whose walk is picks[batch_size + tries]
            res.add(picks[batch_size + tries])
            tries += 1
            if tries > 20:
                break
        while len(res) < batch_size:
            res.add(random.randint(0, n - 1))
        return res

    def sample(self, batch_size):
        return [self._storage[i] for i in self._sample_proportional(batch_size)]

    # ---- inspection (tests, CSV dumps): device -> host --------------------------------------------------------------
    def get_max_priority(self):
        return float(self.state[2].item())

    def get_min_priority(self):
        return float(self.state[3].item())

    def dump_priorities(self, vertex_list):
        k2i = self._key_to_idx
        idx = torch.as_tensor([k2i[v] for v in vertex_list], dtype=torch.int64).to(self.device)
        return self.node[self.cap + idx].cpu().tolist()

    def to_host(self):
        """An equivalent host ``PrioritizedReplayBuffer`` (same leaves, same running extrema): the reference-semantics object
        tests compare against."""
        from .replay_buffer import PrioritizedReplayBuffer
        h = PrioritizedReplayBuffer(self._maxsize, self._alpha, self._max_clip_priority, self._min_clip_priority)
        n = len(self._storage)
        h._storage = list(self._storage)
        h._next_idx = n
        if n:
            h._note_keys(self._storage, np.arange(n))
            h._it_sum.set_many(np.arange(n), self.node[self.cap:self.cap + n].cpu().numpy())
        st = self.state.cpu().tolist()
        for name, val in zip(("_max_priority", "_min_priority", "max_val", "min_val"), st):
            # the host class keeps the untouched sentinels as ints
            setattr(h, name, int(val) if val in (-1.0, 99999999.0) else val)
        return h
