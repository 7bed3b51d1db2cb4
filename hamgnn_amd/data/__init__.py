from .graph import Graph, collate  # noqa: F401
