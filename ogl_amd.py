"""Import alias: the product package lives in the directory ``online-gnn-learning_amd/`` (the name
the build contract fixes); a hyphen is not a Python identifier, so this module loads that
directory as the package ``ogl_amd``.  ``import ogl_amd`` / ``from ogl_amd.graphsage import ...``.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "online-gnn-learning_amd")
_spec = importlib.util.spec_from_file_location(
    "ogl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ogl_amd"] = _mod
_spec.loader.exec_module(_mod)
