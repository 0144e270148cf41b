"""Import alias: ``import mini_sglang_amd`` loads the package kept in ``mini-sglang_amd/``.

The directory name carries a hyphen (it is the project's name), which Python cannot import
directly; this module turns itself into that package by pointing ``__path__`` at the
directory and running its ``__init__``.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mini-sglang_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"), globals())
del _f, _os
