import os
import sys

import pytest

# the tests read the committed golden library directories: no library caches are written next to those (the cache has its
# own tests, which switch it on for directories of their own)
os.environ.setdefault("MIRGE_LIB_CACHE", "0")

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
