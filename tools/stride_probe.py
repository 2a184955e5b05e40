#!/usr/bin/env python3
"""Does the plane stride matter to K1?  The Match planes of an engine lie `stride` = round_up(G, 4096) elements
apart, so at G = 2^24 every plane starts a power of two (128 MiB) after the one before.  Times rafthip_commit at
G = 2^24 + k * 4096 groups (planes k * 32 KiB off the power of two) for V = 5 and 7 voters; planes hold zeros
(the pass is memory-bound, the values do not matter).  A development probe; prints one line per case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import raft_amd

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stream = torch.cuda.Stream(device=dev)
    for V in (5, 7):
        for k in (0, 1, 2, 3, 4, 7, 8, 16, 33, 64, 129, 0):
            G = (1 << 24) + k * 4096
            with raft_amd.Engine(G, V, stream=stream.cuda_stream) as e:
                e.set_uniform_config(V)
                with torch.cuda.stream(stream):
                    for _ in range(3):
                        e.commit()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(20):
                        e.commit()
                    e1.record(stream)
                    e1.synchronize()
                    us = 1e3 * e0.elapsed_time(e1) / 20
            gbs = G * (8 * V + 8) / us / 1e3
            print("V=%d k=%3d G=%d  %.1f us  %.0f GB/s  %.1f %%" % (V, k, G, us, gbs, gbs / 80.0), flush=True)


if __name__ == "__main__":
    main()
