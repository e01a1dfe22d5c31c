#!/usr/bin/env python3
"""Drop-in entry point: same flags as the reference's train.py."""
from infinite_texture_gans_amd.train import main

if __name__ == "__main__":
    main()
