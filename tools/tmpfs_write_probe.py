#!/usr/bin/env python3
"""How fast can the save path put a batch's feature matrix into a page-cache / tmpfs file on this host?  (No GPU needed.)
One writer per FILE is bounded by one thread's copy + page instantiation under the inode's locks; this probe times, for 19.2 MB batches
(60 cuts x 1000 frames x 80 float32): one write() per batch; K threads pwrite-ing disjoint ranges of ONE file; K threads each appending to
its OWN file (stripes); K threads copying into an mmap of one file.  Prints one JSON line.

    python tools/tmpfs_write_probe.py [dir=/dev/shm] [batches=40]"""
import json
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

root = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N = 19_200_000
buf = np.random.rand(N // 4).astype(np.float32)
mv = memoryview(buf).cast("B")
res = {"dir": root, "batch_bytes": N, "batches": NB, "cpus": os.cpu_count()}


def path(k=0):
    return os.path.join(root, f"hipfeat_probe_{os.getpid()}_{k}.bin")


def rate(dt):
    return round(NB * N / dt / 1e9, 2)


fd = os.open(path(), os.O_CREAT | os.O_RDWR | os.O_TRUNC)
t = time.perf_counter()
for i in range(NB):
    os.write(fd, mv)
res["one_file_one_write_GBps"] = rate(time.perf_counter() - t)
os.close(fd)
os.unlink(path())

for K in (2, 4, 8):
    pool = ThreadPoolExecutor(K)
    fd = os.open(path(), os.O_CREAT | os.O_RDWR | os.O_TRUNC)
    t = time.perf_counter()
    for i in range(NB):
        fs = [pool.submit(os.pwrite, fd, mv[k * N // K : (k + 1) * N // K], i * N + k * N // K) for k in range(K)]
        [f.result() for f in fs]
    res[f"one_file_{K}_pwrite_threads_GBps"] = rate(time.perf_counter() - t)
    os.close(fd)
    os.unlink(path())
    fds = [os.open(path(k), os.O_CREAT | os.O_RDWR | os.O_TRUNC) for k in range(K)]
    t = time.perf_counter()
    for i in range(NB):
        fs = [pool.submit(os.write, fds[k], mv[k * N // K : (k + 1) * N // K]) for k in range(K)]
        [f.result() for f in fs]
    res[f"{K}_files_one_writer_each_GBps"] = rate(time.perf_counter() - t)
    for k in range(K):
        os.close(fds[k])
        os.unlink(path(k))
    fd = os.open(path(), os.O_CREAT | os.O_RDWR | os.O_TRUNC)
    os.ftruncate(fd, NB * N)
    m = mmap.mmap(fd, NB * N)
    dst = np.frombuffer(m, dtype=np.uint8)
    src = np.frombuffer(mv, dtype=np.uint8)

    def cp(a, b, off):
        dst[off + a : off + b] = src[a:b]

    t = time.perf_counter()
    for i in range(NB):
        fs = [pool.submit(cp, k * N // K, (k + 1) * N // K, i * N) for k in range(K)]
        [f.result() for f in fs]
    res[f"one_file_mmap_{K}_copy_threads_GBps"] = rate(time.perf_counter() - t)
    del dst
    m.close()
    os.close(fd)
    os.unlink(path())
    pool.shutdown()
print(json.dumps(res))
