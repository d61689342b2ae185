"""Importable alias for the on-disk package directory `jarvis-hybridnet_amd/`.

A hyphen is not legal in a Python module name, so this shim gives the package
an importable name and points its search path at the real directory.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..",
                      "jarvis-hybridnet_amd")
__path__ = [_os.path.normpath(_real)]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
