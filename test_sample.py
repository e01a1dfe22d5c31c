#!/usr/bin/env python3
"""Drop-in entry point: same flags as the reference's test_sample.py."""
from infinite_texture_gans_amd.test_sample import main

if __name__ == "__main__":
    main()
