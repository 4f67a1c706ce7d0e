registry = {}


def register(id, entry_point=None, kwargs=None, **other):
    if id in registry:
        raise ValueError(f"Cannot re-register id: {id}")
    registry[id] = (entry_point, dict(kwargs or {}))
