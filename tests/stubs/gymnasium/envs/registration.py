registry = {}
vector_registry = {}


def register(id, entry_point=None, kwargs=None, vector_entry_point=None, **other):
    if id in registry:
        raise ValueError(f"Cannot re-register id: {id}")
    registry[id] = (entry_point, dict(kwargs or {}))
    if vector_entry_point is not None:
        vector_registry[id] = vector_entry_point
