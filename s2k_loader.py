"""Imports the host package, whose directory name (`rust-seq2kminmers_amd`, as the task names it)
is not a valid Python identifier, under the module name `rust_seq2kminmers_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_NAME = "rust_seq2kminmers_amd"


def import_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    path = os.path.join(_ROOT, "rust-seq2kminmers_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(_NAME, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
