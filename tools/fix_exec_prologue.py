#!/usr/bin/env python3
"""Command-line front end of grl_amd/_exec_prologue.py (the assembly filter the build applies): in.s out.s"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grl_amd._exec_prologue import main

if __name__ == "__main__":
    sys.exit(main(sys.argv))
