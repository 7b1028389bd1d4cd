#!/usr/bin/env python3
"""`python gsplat/examples/simple_trainer.py {default|mcmc} ...` as the reference launches it
(source/container/src/main.py:1339-1347), served by mi3dgs."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.realpath(__file__)), "..", "..", ".."))
from mi3dgs.cli import main_simple_trainer  # noqa: E402

if __name__ == "__main__":
    sys.exit(main_simple_trainer())
