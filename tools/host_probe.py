#!/usr/bin/env python3
"""Host-side facts behind the render loop's numbers (GPU box): CPU quota of the container, memcpy rates between pageable / pinned tensors with the GPU idle and busy."""
import os, sys, time, threading
import torch
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a", "| nproc", os.cpu_count(), "| affinity", len(os.sched_getaffinity(0)), "| torch threads", torch.get_num_threads())
print("loadavg:", open("/proc/loadavg").read().strip())
dev = torch.device("cuda:0")
x = torch.randn(4096, 4096, device=dev)
def busy(n):
    for _ in range(n): torch.mm(x, x)
N = 64
src_pin = [torch.randn(1, 512, 512).pin_memory() for _ in range(N)]
src_pag = [torch.randn(1, 512, 512) for _ in range(N)]
dst_pin = torch.empty(1, 512, 512).pin_memory(); dst_pag = torch.empty(1, 512, 512)
def rate(srcs, dst):
    t0 = time.perf_counter()
    for s in srcs: dst.copy_(s)
    dt = time.perf_counter() - t0
    return N * 1.0 / dt / 1024       # GiB/s (1 MiB each)
for gpu in ("idle", "busy"):
    if gpu == "busy": busy(400)           # ~ a second of queued GEMMs
    print("GPU %s: copy_ of 1-MiB tensors, GiB/s: pageable->pageable %.2f | pinned->pageable %.2f | pageable->pinned %.2f | pinned->pinned %.2f" % (
        gpu, rate(src_pag, dst_pag), rate(src_pin, dst_pag), rate(src_pag, dst_pin), rate(src_pin, dst_pin)))
    torch.cuda.synchronize()
import numpy as np
a = [np.random.rand(262144).astype(np.float32) for _ in range(N)]; d = np.empty(262144, np.float32)
t0 = time.perf_counter()
for s in a: np.copyto(d, s)
print("numpy copyto 1 MiB: %.2f GiB/s" % (N / (time.perf_counter() - t0) / 1024))
