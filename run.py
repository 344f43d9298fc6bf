#!/usr/bin/env python3
"""Repo-root launcher with the reference's run.py flag surface: `python run.py --benchmark -m facebook/opt-30b ...`."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "isca-2025-lia_amd"))
from lia_amd.run import main  # noqa: E402

if __name__ == "__main__":
    main(sys.argv[1:])
