"""TEST INFRASTRUCTURE (loaded by pytest.ini: -p tests.autoparallel).  The CPU suite -- the kernels compiled for the host and run on
fibers, tests/hipemu -- is CPU-bound and its modules are independent: without a GPU in the machine and without an explicit -n, run it on
pytest-xdist workers (6 at most; ~5 minutes instead of ~20 on 8 cores).  On a GPU box (/dev/kfd present) nothing changes: one process
owns the device, as the driver's `pytest -m gpu` expects."""
import os


def pytest_load_initial_conftests(early_config, parser, args):
    if os.path.exists("/dev/kfd") or os.environ.get("SA_TEST_SERIAL"):
        return
    for a in args:
        if a == "-n" or a.startswith("-n") or a.startswith("--numprocesses") or a.startswith("--dist") or a in ("--collect-only", "--co", "--pdb"):
            return
    try:
        import xdist  # noqa: F401
    except ImportError:
        return
    n = min(6, os.cpu_count() or 1)
    if n > 1:
        args[:] = list(args) + ["-n", str(n)]
