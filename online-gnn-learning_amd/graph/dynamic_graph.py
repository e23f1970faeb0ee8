"""Dynamic-graph interface (same surface as R/train/graph/dynamic_graph.py:9-43)."""


class DynamicGraph:
    def __init__(self, graph, snapshots, labelled_vertices, search_depth):
        assert snapshots > 0
        self.graph = graph
        self.snapshots = snapshots
        self.search_depth = search_depth
        self.evolution_index = 0
        self.labelled_vertices = labelled_vertices

    def get_labelled_vertices(self):
        return self.labelled_vertices

    def get_added_vertices(self):
        raise NotImplementedError

    def get_graph(self):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError

    def evolve(self):
        raise NotImplementedError
