#!/usr/bin/env python3
"""`training/colmap_to_nerfstudio_cam.py -d D` as the reference runs it (source/container/src/main.py:1220-1226)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.realpath(__file__)), "..", ".."))
from mi3dgs.colmap_json import main  # noqa: E402

sys.exit(main())
