"""TEST INFRASTRUCTURE: builds and binds the host-emulated kernel library (tests/hipemu).

The emulated library compiles the unmodified searcharray_amd/csrc/*.hip sources with g++ against
a fiber-based HIP stand-in so kernel LOGIC is checked on machines without a GPU.  It is never
used by the package itself (searcharray_amd._lib only loads the gfx950 build).
"""
import ctypes
import fcntl
import os
import subprocess

from searcharray_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "_emu", "libsearcharray_emu.so")
_api = None


def emu_api():
    global _api
    if _api is None:
        so = os.environ.get("SA_EMU_SO")                # e.g. the same sources built with -fsanitize=address (run pytest under
        if not so:                                      #      LD_PRELOAD=libasan.so: every kernel access becomes a checked access)
            # (one builder at a time: pytest-xdist workers start together, and a worker must not load a library another is still linking)
            os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
            with open(os.path.join(os.path.dirname(EMU_SO), ".build.lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "searcharray_amd", "csrc"), "emu"])
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
            so = EMU_SO
        _api = _lib.bind(ctypes.CDLL(so), so)
    return _api
