#!/usr/bin/env python3
"""Many forks: a chain of 300 small launches on the capture stream; after every 5th one the side stream waits for the
chain so far and gets two launches of its own (the shape of a backward pass whose weight gradients fork per layer);
one join at the end.  Eager vs graph replay, with the side work on the side stream vs inline on the chain's stream."""
import time

import torch

dev = "cuda"
a = torch.randn(512, 512, device=dev)
b = torch.randn(512, 512, device=dev)
side = torch.cuda.Stream()
ev = torch.cuda.Event()


EVERY = 5


def work(fork):
    cur = torch.cuda.current_stream()
    x, outs, pend = a, [], []
    for i in range(300):
        x = x @ a * 0.01
        if i % 5 == 4:
            pend.append(x)
        if i % EVERY == EVERY - 1 and pend:
            if fork:
                ev.record()
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    for t in pend:
                        outs.append((t @ b) @ b)
            else:
                for t in pend:
                    outs.append((t @ b) @ b)
            pend = []
    if fork:
        cur.wait_stream(side)
    return x, outs


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


import sys
for EVERY, fork in ((5, False), (5, True), (25, True), (75, True), (150, True)):
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
    print(f"fork={fork} (side work flushed every {EVERY} launches): eager {eager:.2f} ms, graph replay {rep:.2f} ms")
