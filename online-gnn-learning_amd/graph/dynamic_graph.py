"""Abstract stream-of-snapshots interface shared by the vertex- and edge-stream graphs.

API parity target: the base class in R/train/graph/dynamic_graph.py:9-43 (attributes ``graph``, ``snapshots``,
``search_depth``, ``evolution_index``, ``labelled_vertices`` and the five overridable methods).  Implemented as an
ABC so a subclass that forgets part of the surface fails at construction instead of at the first snapshot.
"""
import abc


class DynamicGraph(abc.ABC):
    def __init__(self, graph, snapshots, labelled_vertices, search_depth):
        if snapshots <= 0:
            raise AssertionError("a stream needs at least one snapshot")
        self.graph, self.snapshots = graph, snapshots
        self.labelled_vertices = labelled_vertices      # ids whose target is known (label != -1)
        self.search_depth = search_depth                # hops around an update that count as "changed"
        self.evolution_index = 0                        # number of snapshots applied so far

    def get_labelled_vertices(self):
        return self.labelled_vertices

    @abc.abstractmethod
    def get_added_vertices(self, delta=None):
        """(vertices added by the last ``delta`` snapshots, parallel list of "is labelled" flags)."""

    @abc.abstractmethod
    def get_graph(self):
        """The snapshot graph the sampler reads."""

    @abc.abstractmethod
    def __len__(self):
        """Number of snapshots in the stream."""

    @abc.abstractmethod
    def evolve(self):
        """Apply the next snapshot."""
