#!/usr/bin/env python3
"""`post_processing/rotate_splat.py` as the reference launches it (source/container/src/main.py:1481-1523, 1556-1592).
Default output equals the reference script's; `--sh-mode exact` rotates every SH band properly."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.realpath(__file__)), "..", ".."))
from mi3dgs.transform import main_rotate  # noqa: E402

sys.exit(main_rotate())
