"""MEASUREMENT-SCRIPT HELPER (not part of the package): the scripts in this directory predate the options struct of the C
ABI (include/searcharray_hip.h, Part 0) and switch library routes by writing SA_* environment variables.  The library no
longer reads those.  Importing this module installs a thin proxy over ``os.environ`` that mirrors every SA_<OPTION> write,
pop or update into the calling thread's scoped options (searcharray_amd.options.Scope), so the scripts' command lines
(``SA_SPARSE=0 python scripts/x.py``) and their in-process toggles keep meaning what they meant."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from searcharray_amd import options as _options  # noqa: E402

_scope = _options.Scope()
_ALIASES = {"phrase_trace": "trace", "span_trace": "trace"}


def _known(key: str):
    if not isinstance(key, str) or not key.startswith("SA_"):
        return None
    name = _ALIASES.get(key[3:].lower(), key[3:].lower())
    # (the option names are only known once a library is bound: accept any SA_ name whose lower-case form looks like one)
    return name if name in _NAMES else None


_NAMES = set()
with open(os.path.join(ROOT, "searcharray_amd", "csrc", "sa_options.hpp")) as _f:
    import re as _re
    _NAMES = set(_re.findall(r"^\s+X\((\w+)\)", _f.read(), _re.M))


class _Env:
    def __init__(self, real):
        object.__setattr__(self, "_real", real)

    def __getattr__(self, a):
        return getattr(self._real, a)

    def __getitem__(self, k):
        return self._real[k]

    def __contains__(self, k):
        return k in self._real

    def __iter__(self):
        return iter(self._real)

    def __len__(self):
        return len(self._real)

    def __setitem__(self, k, v):
        self._real[k] = v
        n = _known(k)
        if n:
            _scope.set(n, v)

    def __delitem__(self, k):
        del self._real[k]
        n = _known(k)
        if n:
            _scope.unset(n)

    def pop(self, k, *d):
        n = _known(k)
        if n:
            _scope.unset(n)
        return self._real.pop(k, *d)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def get(self, k, d=None):
        return self._real.get(k, d)

    def copy(self):
        return self._real.copy()

    def keys(self):
        return self._real.keys()

    def items(self):
        return self._real.items()


if not isinstance(os.environ, _Env):
    for _k, _v in list(os.environ.items()):
        _n = _known(_k)
        if _n:
            _scope.set(_n, _v)
    os.environ = _Env(os.environ)
