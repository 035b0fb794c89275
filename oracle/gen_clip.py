#!/usr/bin/env python3
"""Command-line shim kept for the scripts: the synthetic clip generator lives in thor_amd/synth.py (it is the
bench's input generator, not part of the oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thor_amd.synth import *  # noqa: F401,F403
from thor_amd.synth import main

if __name__ == '__main__':
    main()
