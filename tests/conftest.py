"""pytest configuration: markers, library builds, shared fixtures.

`-m "not gpu"`: oracle vs the reference's KATs, host logic, C-ABI symbol checks (no compute).
`-m gpu`:       parity of the HIP path against the oracle, through the C ABI, on a real MI355X.
"""
import importlib
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A hung kernel must not sit there silently: pytest-timeout (when installed) dumps the Python stacks and
    # ends the process, so the test that hung is named in the log.
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(240, method="thread"))


def _built():
    need = ["ray-tracer-challenge_amd/lib/librtc_hip.so", "ray-tracer-challenge_amd/lib/librtc_host.so",
            "ray-tracer-challenge_amd/lib/rtc_host_kat", "oracle/build/liboracle.so", "oracle/build/oracle_kat"]
    return all(os.path.exists(os.path.join(REPO, p)) for p in need)


@pytest.fixture(scope="session", autouse=True)
def build_everything():
    """Builds the libraries once per session if they are missing (the GPU box gets them prebuilt)."""
    if not _built():
        subprocess.run(["make", "-C", REPO, "all"], check=True, stdout=subprocess.DEVNULL)
    yield


@pytest.fixture(scope="session")
def rtc():
    mod = importlib.import_module("ray-tracer-challenge_amd")
    # On a GPU box, let PyTorch bring up its HIP context BEFORE this library launches anything (the order bench.py
    # and smoke() have always used).  Twice a GPU test run hung at the first torch.cuda call made after the
    # library had already rendered in the same process; the cause was never found, this order never hung.
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
            torch.cuda.synchronize()
    except ImportError:
        pass
    return mod
