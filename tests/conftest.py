import os
import sys

import pytest

# A plan serves batches below ~2^23 samples with the full-recurrence kernel
# (no per-launch seed table to build: cordic_kernels.hip: seed_min_samples).
# The tests run the table-seeded kernels at SMALL sizes on purpose, and so do
# the bench.py runs they start: always seeded here.  (Read once by the library,
# at its first launch.)
os.environ.setdefault("CORDIC_SEED_MIN_SAMPLES", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden",
                           "gencordic_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle and the product library exist (build() is cheap
    when everything is up to date; on the GPU box the prebuilt .so files that
    travelled with the snapshot are used as they are)."""
    import __graft_entry__ as g
    g.build()
