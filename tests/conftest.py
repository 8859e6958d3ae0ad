import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "software-rasterizer_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure), built on demand, spot texture registered in slot 0."""
    from oracle import oracle
    import scenes
    oracle.lib()
    oracle.texture_set(scenes.TEX_SPOT, scenes.spot_texture())
    return oracle
