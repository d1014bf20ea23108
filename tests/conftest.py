"""pytest configuration: `gpu` marker + repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer-running CPU test")
    config.addinivalue_line("markers", "perf: timing guard, needs an idle MI355X (pytest -m perf); not part of -m gpu")
