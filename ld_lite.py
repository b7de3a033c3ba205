#!/usr/bin/env python3
"""ld_lite with the reference's command line (see ld_tools_amd/cli.py); the LD arithmetic runs on the MI355X."""
from ld_tools_amd.cli import ld_lite_main

if __name__ == "__main__":
    ld_lite_main()
