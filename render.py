#!/usr/bin/env python3
"""Drop-in for AdaPT's `render.py` (pt renderer only):  python render.py --scene cbox --name c2_cbox.xml --iter_num 128 --no_gui"""
import sys

from adapt_amd.cli import main

if __name__ == "__main__":
    sys.exit(main())
