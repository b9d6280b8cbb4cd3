#!/usr/bin/env python3
"""Does a HIP-graph replay run forked branches concurrently?  Two chains of 300 small dependent launches (each far too
small to fill the chip), one on the capture stream and one on a forked stream, joined at the end: eager issue vs graph
replay (ms), forked vs both on one stream.  If the forked replay takes about as long as the one-stream replay, the
executor serialises the branches (or orders them level by level)."""
import time

import torch

dev = "cuda"
a = torch.randn(512, 512, device=dev)
b = torch.randn(512, 512, device=dev)
side = torch.cuda.Stream()


def work(fork):
    cur = torch.cuda.current_stream()
    def chain(m):
        y = m
        for _ in range(300):
            y = y @ m * 0.01
        return y
    if fork:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            y = chain(b)
    else:
        y = chain(b)
    x = a
    for _ in range(300):
        x = x @ a * 0.01
    if fork:
        cur.wait_stream(side)
    return x, y


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for fork in (False, True):
    eager = timed(lambda: work(fork))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        work(fork)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        work(fork)
    rep = timed(g.replay)
    print(f"fork={fork}: eager {eager:.2f} ms, graph replay {rep:.2f} ms")
