class HipExecutable:
    def __init__(self, graph):
        self.graph = graph
    def __call__(self, *a):
        raise RuntimeError("stub")
