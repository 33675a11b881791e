#!/usr/bin/env python3
"""How much does the chip gain from running two INDEPENDENT hydro pipelines at once?  Two Sedov problems of n^3, each with
its own scratch context, stepped (a) one after the other on one stream and (b) from two host threads on two streams, the
second started half a step late so that unlike kernels meet.  (b)/(a) < 1 is what a slab-pipelined single problem could
gain at best from overlapping its arithmetic-bound kernels (k_trace) with its byte-bound ones.
usage: tools/concurrency_probe.py [n] [steps]"""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
import castro_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def make():
    c = castro_amd.Castro((n, n, n))
    c.initData("sedov")
    for _ in range(3):
        c.step()
    return c


A, B = make(), make()
torch.cuda.synchronize()


def run(c, stream, delay, out, k):
    with torch.cuda.stream(stream):
        time.sleep(delay)
        t0 = time.perf_counter()
        for _ in range(steps):
            c.step()
        stream.synchronize()
        out[k] = time.perf_counter() - t0


for rep in range(2):
    t0 = time.perf_counter()
    for c in (A, B):
        for _ in range(steps):
            c.step()
    torch.cuda.synchronize()
    seq = time.perf_counter() - t0

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    out = [0.0, 0.0]
    step_s = seq / (2 * steps)
    ta = threading.Thread(target=run, args=(A, sa, 0.0, out, 0))
    tb = threading.Thread(target=run, args=(B, sb, 0.5 * step_s, out, 1))
    t0 = time.perf_counter()
    ta.start(); tb.start(); ta.join(); tb.join()
    torch.cuda.synchronize()
    con = time.perf_counter() - t0
    print("n=%d steps=%d: sequential %.2f ms per step-pair, concurrent %.2f ms per step-pair (ratio %.3f); threads took %.1f / %.1f ms per step"
          % (n, steps, 1e3 * seq / steps, 1e3 * con / steps, con / seq, 1e3 * out[0] / steps, 1e3 * out[1] / steps))
