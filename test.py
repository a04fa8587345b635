#!/usr/bin/env python3
"""python test.py --config=<yaml>   (same entry point as the reference's test.py)"""
from vae_conditional_sampling_amd.driver import main

if __name__ == "__main__":
    main()
