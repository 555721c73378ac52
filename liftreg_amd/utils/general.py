"""Plugin loader with the reference's semantics (src/liftreg/utils/general.py:9-15)."""
import importlib


def get_class(kls):
    """'a.b.C' → attribute C of module a.b (any importable dotted path)."""
    module_name, _, attr = kls.rpartition(".")
    if not module_name:
        raise ValueError(f"'{kls}' is not a dotted class path")
    return getattr(importlib.import_module(module_name), attr)
