#!/usr/bin/env python3
"""Repeat the conv parity cases many times in one process (race / stale-memory screen)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_ops as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = 0
for it in range(n):
    for case in T.CONV_CASES:
        # poison the allocator's free blocks so stale reads show up
        junk = torch.full((64 << 20,), float("nan"), device="cuda"); del junk
        try:
            T.test_conv_fwd_dgrad_wgrad(case)
        except AssertionError as e:
            bad += 1
            print("FAIL iter", it, case[0], str(e)[:200].replace("\n", " "), flush=True)
print("failures:", bad)
