"""Drop-in module path of the reference package (`sys.path.insert(0, '<repo>/python')`,
examples/example.py:10-12).  The implementation lives in demon_amd."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "..")))
