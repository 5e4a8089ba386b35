import importlib

from .. import error

registry = {}


def register(id, entry_point=None, **kwargs):
    registry[id] = (entry_point, kwargs)


def make(id, **kwargs):
    if id not in registry:
        raise error.UnregisteredEnv("No registered env with id: {}".format(id))
    entry_point, reg_kwargs = registry[id]
    if callable(entry_point):
        cls = entry_point
    else:
        mod_name, attr = entry_point.split(':')
        cls = getattr(importlib.import_module(mod_name), attr)
    kw = dict(reg_kwargs.get('kwargs', {}))
    kw.update(kwargs)
    return cls(**kw)
