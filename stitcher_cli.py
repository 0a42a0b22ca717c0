#!/usr/bin/env python3
"""`python stitcher_cli.py -i DIR -r -ff ...` -- the reference's entry point name, forwarding to the
MI355X drop-in (image-stitcher_amd/stitcher_cli.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from image_stitcher_amd.stitcher_cli import main  # noqa: E402

if __name__ == '__main__':
    main()
