"""The library's switches as data (include/searcharray_hip.h, Part 0): one ``sa_options_t`` per index handle and per
batch -- the counterpart of the keyword arguments of the reference's objects (searcharray/postings.py:250-258, :652-656).

``Options(**kw)`` holds named values (anything not named is left to the library); ``DeviceIndex(..., opts=...)``,
``dev.batch(..., opts=...)``, ``obj.set_options(**kw)`` hand them to the C ABI.  ``scoped(**kw)`` is a context manager for
code that cannot thread an argument through (benchmarks, tests): inside it, every handle THIS thread creates or uses takes
the given values on top of its own -- a thread-local, so two threads of a process can hold different settings.

The field list is read from the library itself (``sa_option_count`` / ``sa_option_name``), so this module cannot drift
from the header."""
from __future__ import annotations

import contextlib
import ctypes
import threading
from typing import Dict, Optional

UNSET = -(1 << 63)                       # SA_OPT_UNSET
_names_by_api: Dict[int, list] = {}
_tls = threading.local()


def names(api) -> list:
    key = id(api)
    if key not in _names_by_api:
        if not hasattr(api, "sa_option_count"):          # (scripts/ab.py binding a build from before Part 0 existed: no options)
            _names_by_api[key] = []
        else:
            n = api.sa_option_count()
            _names_by_api[key] = [api.sa_option_name(i).decode() for i in range(n)]
    return _names_by_api[key]


def supported(api) -> bool:
    return bool(names(api))


def _norm(name: str) -> str:
    """'SA_SPARSE' / 'sparse' -> 'sparse' (the switches' former environment names are accepted)"""
    n = name.lower()
    return n[3:] if n.startswith("sa_") else n


_PHRASE_MODES = {"auto": 0, "general": 1, "fused": 2}


def _value(name: str, v) -> int:
    if name == "phrase_mode" and isinstance(v, str) and not v.lstrip("-").isdigit():
        return _PHRASE_MODES[v]
    return int(v)


class Options(dict):
    """name -> int64; missing = the library decides"""

    def __init__(self, *a, **kw):
        super().__init__()
        for d in a:
            if d:
                self.update_from(d)
        self.update_from(kw)

    def update_from(self, d) -> "Options":
        for k, v in dict(d).items():
            k = _norm(k)
            if v is None:
                self.pop(k, None)
            else:
                self[k] = _value(k, v)
        return self

    def merged(self, other) -> "Options":
        return Options(self, other)

    def struct(self, api):
        """a filled sa_options_t (ctypes buffer: struct_size + one int64 per field, in the library's order)"""
        fields = names(api)
        buf = (ctypes.c_int64 * (1 + len(fields)))()
        api.sa_options_init(ctypes.byref(buf))
        for k, v in self.items():
            if k not in fields:
                raise KeyError(f"unknown option '{k}' (known: {', '.join(fields)})")
            buf[1 + fields.index(k)] = v
        return buf


def ambient() -> Options:
    """the calling thread's scoped options (empty outside ``scoped``)"""
    return getattr(_tls, "opts", None) or Options()


def ambient_version() -> int:
    return getattr(_tls, "version", 0)


def _set_ambient(o: Options) -> None:
    _tls.opts = o
    _tls.version = getattr(_tls, "version", 0) + 1


@contextlib.contextmanager
def scoped(**kw):
    old = ambient()
    _set_ambient(old.merged(kw))
    try:
        yield
    finally:
        _set_ambient(old)


class Scope:
    """imperative form of ``scoped`` for test fixtures: set / unset until ``close``"""

    def __init__(self):
        self._saved = ambient()

    def set(self, name=None, value=None, **kw):
        d = dict(kw)
        if name is not None:
            d[name] = value
        _set_ambient(ambient().merged(d))

    def unset(self, *names_):
        o = Options(ambient())
        for n in names_:
            o.pop(_norm(n), None)
        _set_ambient(o)

    def close(self):
        _set_ambient(self._saved)


@contextlib.contextmanager
def creating(api, opts: Optional[Options]):
    """handles created inside start from ``opts`` + the thread's scoped options (sa_options_set_thread_defaults)"""
    o = Options(opts, ambient())
    if not supported(api):
        o = Options()
    if o:
        buf = o.struct(api)
        api.call("sa_options_set_thread_defaults", ctypes.byref(buf))
    try:
        yield o
    finally:
        if o:
            api.call("sa_options_set_thread_defaults", None)


class OptionsMixin:
    """for objects that wrap an index / batch handle: ``self._h`` the handle, ``self.api``, ``self._opt_setter`` the C
    function that replaces the handle's options.  The handle follows the thread's scoped options: ``_sync_opts`` (called
    at the top of every method that uses the handle) re-applies base + scoped values when the scope has changed."""
    _opt_setter = ""

    def _init_opts(self, base: Optional[Options]) -> None:
        self._opts_base = Options(base)
        self._opts_seen = (ambient_version(), threading.get_ident())
        self._opts_applied = Options(self._opts_base, ambient())

    def _apply_opts(self) -> None:
        o = Options(self._opts_base, ambient())
        if o != self._opts_applied and supported(self.api):
            buf = o.struct(self.api)
            self.api.call(self._opt_setter, self._h, ctypes.byref(buf))
            self._opts_applied = o
        self._opts_seen = (ambient_version(), threading.get_ident())

    def _sync_opts(self) -> None:
        if self._opts_seen != (ambient_version(), threading.get_ident()):
            self._apply_opts()

    def _call(self, name: str, *args) -> None:
        """a C-ABI call on this handle, under the thread's current options"""
        self._sync_opts()
        self.api.call(name, *args)

    def set_options(self, **kw) -> None:
        """replace named switches of this handle (None: back to the library's choice)"""
        self._opts_base.update_from(kw)
        self._apply_opts()

    def options(self) -> Options:
        return Options(self._opts_applied)
