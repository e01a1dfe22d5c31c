#!/usr/bin/env python3
"""End-to-end throughput of the train.py command line at config 1's flags on a synthetic texture image (usage, GPU box:
python tools/cli_throughput.py): the real-image pipeline (decode once, random crops, host -> device copy, CPU latents)
in front of the same step bench.py times on resident synthetic inputs."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import bench  # noqa: E402
from infinite_texture_gans_amd import train as T  # noqa: E402

sampling = int(sys.argv[1]) if len(sys.argv) > 1 else 2400          # crops per epoch -> 300 iterations at batch 8
with tempfile.TemporaryDirectory() as d:
    rng = np.random.RandomState(0)
    Image.fromarray(rng.randint(0, 255, (768, 1024, 3), dtype=np.uint8)).save(os.path.join(d, "tex.jpg"))
    flags = [f for f in bench.FLAGS]
    argv = flags + ["--data_path", os.path.join(d, "tex.jpg"), "--sampling", str(sampling), "--epochs", "3",
                    "--fname", os.path.join(d, "cp")] + sys.argv[2:]          # e.g. --launch_mode eager
    t = []
    real_print = print

    import builtins

    def tap(*a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("["):
            t.append(time.perf_counter())
        real_print(*[str(x)[:160] for x in a][:1], **k) if a and isinstance(a[0], str) and a[0].startswith("[") else None
    builtins.print = tap
    try:
        T.main(argv)
    finally:
        builtins.print = real_print
    torch.cuda.synchronize()
    its = sampling // 8
    for i in range(1, len(t)):
        print("epoch %d: %d iterations in %.3f s -> %.1f crops/s" % (i + 1, its, t[i] - t[i - 1], sampling / (t[i] - t[i - 1])))
