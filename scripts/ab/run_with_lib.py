#!/usr/bin/env python
"""Same-box A/B of two builds of the C-ABI library: python scripts/ab/run_with_lib.py <lib.so> <script.py> [args...] runs the
script with lora_amd._C bound to that library (measurement only; the product loads lora_amd/csrc/liblora_amd.so)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lora_amd import _C  # noqa: E402

_C.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
