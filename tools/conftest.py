"""sys.path for the dev tools (repo root, the package, tests/ for the oracle-built scenes) — not a pytest conftest."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "software-rasterizer_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
