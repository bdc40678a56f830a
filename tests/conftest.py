import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"),
          os.path.join(ROOT, "g-vom_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


# Collection order: the parity tests proper come first, anything that starts bench.py (and so depends
# on how busy the box is) last -- under `pytest -x` a slow box can then never stop the run in front of a parity test.
_ORDER = ("test_abi", "test_oracle_kat", "test_oracle_golden", "test_ingest_cpu", "test_hip_parity", "test_hip_sharded",
          "test_sanitizers", "test_sharded_gloo", "test_comm_rendezvous", "test_bench_contract")


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else len(_ORDER) - 1
    items.sort(key=rank)                                    # stable: the order inside a file stays


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
