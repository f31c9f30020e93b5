#!/usr/bin/env python3
"""Does a captured hipGraph run forked branches concurrently?  Branch A = one HBM-bound pass,
branch B = a chain of small latency-bound kernels (a stand-in for Adam vs the binning stage)."""
import torch

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
big = torch.rand(160_000_000, device=dev)       # 640 MB read+write per pass
small = torch.rand(500_000, device=dev)
side = torch.cuda.Stream()


import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qed_splatter_amd import _lib as L
lib = L.load()
NP = 29_500_000
P, G, M, V = (torch.rand(NP, device=dev) for _ in range(4))
begins = (C.c_int64 * 2)(0, NP)
lrs = (C.c_float * 1)(1e-3)


def branch_a():
    L.check(lib.qed_adam_step(L.ptr(P), L.ptr(G), L.ptr(M), L.ptr(V), 1, begins, lrs, 0.9, 0.999, 1e-15, 5, None,
                              torch.cuda.current_stream().cuda_stream), "adam")


def branch_b():
    for _ in range(40):
        small.add_(1.0)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def capture(fork):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        main = torch.cuda.current_stream()
        if fork:
            ev = torch.cuda.Event(); ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                branch_a()
                ev2 = torch.cuda.Event(); ev2.record(side)
            branch_b()
            main.wait_event(ev2)
        else:
            branch_a(); branch_b()
    return g


for _ in range(3):
    branch_a(); branch_b()
torch.cuda.synchronize()
ga = torch.cuda.CUDAGraph()
with torch.cuda.graph(ga, stream=torch.cuda.current_stream()):
    branch_a()
gb = torch.cuda.CUDAGraph()
with torch.cuda.graph(gb, stream=torch.cuda.current_stream()):
    branch_b()
print("graph A alone   %.1f us" % timed(ga.replay))
print("graph B alone   %.1f us" % timed(gb.replay))
print("graph serial    %.1f us" % timed(capture(False).replay))
print("graph forked    %.1f us" % timed(capture(True).replay))


def eager_fork():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        branch_a()
        ev2 = torch.cuda.Event(); ev2.record(side)
    branch_b()
    main.wait_event(ev2)


print("eager forked    %.1f us" % timed(eager_fork))
print("eager serial    %.1f us" % timed(lambda: (branch_a(), branch_b())))
