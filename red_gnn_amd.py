"""Import alias for the package directory ``red-gnn_amd/``.

The package directory name is fixed by the repo layout contract but is not a legal
Python identifier, so this one-file module gives it the importable name
``red_gnn_amd``: a module that defines ``__path__`` is a package to the import
system, so ``import red_gnn_amd.models`` resolves to ``red-gnn_amd/models.py``.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "red-gnn_amd")]

from red_gnn_amd._pkg import *  # noqa: E402,F401,F403
from red_gnn_amd._pkg import __all__, __version__  # noqa: E402,F401
