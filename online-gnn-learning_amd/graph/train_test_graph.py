"""Train/test split + replay state over a dynamic graph (host-side seed selection).

Same surface and behaviour as R/train/graph/train_test_graph.py:12-248 (SURVEY.md §8(b) "graph-state
surface used by the path"): 85/15 split of newly arrived labelled vertices with
``sklearn.model_selection.train_test_split``; new vertices enter the buffer at
``min + (max - min) * 0.95`` (or ``start_priority`` while the buffer is empty); alpha annealed
linearly per snapshot; ``draw_priority_train_nodes`` is a uniform shuffle prefix whenever
``n <= |train|`` (the reference's behaviour, kept); a full priority pass re-creates the buffer.
"""
from __future__ import annotations

import random
from itertools import compress

from sklearn.model_selection import train_test_split

from ..prioritized_replay.replay_buffer import PrioritizedReplayBuffer

SIZE_BUFFER = 10000000


class TrainTestGraph:
    def __init__(self, graph, split=0.25, start_prior_alpha=1, end_prior_alpha=2, scale=1, max_priority=3.0,
                 start_priority=2, min_priority=0.0000001):
        self.scale = scale
        self.temporal_graph = graph
        self.train_set, self.test_set = set(), set()
        self.size_evolution = len(graph)
        self.split = split
        self.graph = self.temporal_graph.get_graph()
        self.prior_alpha = start_prior_alpha
        self.start_prior_alpha, self.end_prior_alpha = start_prior_alpha, end_prior_alpha
        self.max_priority, self.start_priority, self.min_priority = max_priority, start_priority, min_priority
        self.priority_replay_buffer = self._new_buffer()
        added, labelled = self.temporal_graph.get_added_vertices()
        self._draw_train_test(list(compress(added, labelled)))

    def _new_buffer(self):
        return PrioritizedReplayBuffer(SIZE_BUFFER, self.prior_alpha, max_priority=self.max_priority,
                                       min_priority=self.min_priority)

    def _draw_train_test(self, vertices):
        if len(vertices) >= 3:
            self.train, self.test = train_test_split(vertices, shuffle=True, test_size=self.split)
        else:
            self.train, self.test = set(vertices), set()
        self.train_set = self.train_set.union(set(self.train))
        self.train_set_list = list(self.train_set)
        self.test_set = self.test_set.union(set(self.test))
        self.test_set_list = list(self.test_set)
        self._update_priority_struct()

    def _update_priority_struct(self):
        buf = self.priority_replay_buffer
        if buf.get_max_priority() == -1:
            value = self.start_priority
        else:
            lo, hi = buf.get_min_priority(), buf.get_max_priority()
            value = lo + (hi - lo) * 0.95
        buf.add_all({v: value for v in self.train})

    def __len__(self):
        return len(self.temporal_graph)

    def evolve(self):
        self.prior_alpha = self.start_prior_alpha + (
            ((self.end_prior_alpha - self.start_prior_alpha) / self.__len__()) * self.temporal_graph.evolution_index)
        self.temporal_graph.evolve()
        self.graph = self.temporal_graph.get_graph()
        added, labelled = self.temporal_graph.get_added_vertices()
        self._draw_train_test(list(compress(added, labelled)))

    def get_graph(self):
        return self.temporal_graph.get_graph()

    def get_train_set(self):
        return self.train_set_list

    def get_test_set(self):
        return self.test_set_list

    def get_new_train_nodes(self, batch_size=None):
        l_train = list(self.train)
        if batch_size is None or batch_size >= len(l_train):
            return l_train
        random.shuffle(l_train)
        return l_train[:batch_size]

    def get_new_test_nodes(self):
        return self.test

    def draw_random_train_nodes(self, n_nodes):
        if n_nodes <= len(self.train_set_list):
            random.shuffle(self.train_set_list)
            return self.train_set_list[:n_nodes]
        return self.train_set_list

    def draw_priority_train_nodes(self, n_nodes):
        if n_nodes <= len(self.train_set_list):
            random.shuffle(self.train_set_list)
            return self.train_set_list[:n_nodes]
        return self.priority_replay_buffer.sample(n_nodes)

    def dump_priorities(self, vertex_list):
        return self.priority_replay_buffer.dump_priorities(vertex_list)

    def update_priorities(self, d_priorities):
        assert len(d_priorities) <= len(self.train_set)
        if len(d_priorities) < len(self.train_set):
            self.priority_replay_buffer.update_priorities(d_priorities)
        else:
            self.priority_replay_buffer = self._new_buffer()
            self.priority_replay_buffer.add_all(d_priorities)

    def get_original_to_subgraph_map(self):
        return self.temporal_graph.get_original_to_subgraph_map()

    def get_subgraph_to_original_map(self):
        return self.temporal_graph.get_subgraph_to_original_map()
