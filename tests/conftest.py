import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the host build workers must exist before the first test makes a GPU context (a process with one never forks):
    # every HostScene of the session is then built by them
    from metadrive_ped_amd import hostpool
    hostpool.start()


def pytest_unconfigure(config):
    from metadrive_ped_amd import hostpool
    hostpool.stop()


@pytest.fixture(scope="session")
def cs_dist():
    """Block distribution restricted to the block types built so far (Curve / Straight)."""
    from collections import OrderedDict
    from metadrive_ped_amd.mapgen.pg import BLOCK_TYPE_DISTRIBUTION_V2
    d = OrderedDict((k, 0.0) for k in BLOCK_TYPE_DISTRIBUTION_V2)
    d["Curve"], d["Straight"] = 0.6, 0.4
    return d
