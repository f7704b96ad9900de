"""The device brushfire queue (iris_lama_amd/csrc/lama_heap.h) pops in EXACTLY the order of
std::priority_queue with the reference's comparator, ties included -- the brushfire result depends on it."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib():
    out = os.path.join(HERE, "cpu_engine", "_build", "libheap_shim.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-o", out, os.path.join(HERE, "heap_shim.cpp")], check=True)
    return C.CDLL(out)


def test_heap_matches_libstdcxx_priority_queue_with_ties():
    L = _lib()
    rng = np.random.default_rng(0)
    for trial in range(30):
        n = int(rng.integers(50, 4000))
        span = int(rng.choice([1, 3, 10, 100]))          # few distinct priorities => many ties
        ops = np.where(rng.random(n) < 0.6, rng.integers(0, span, size=n), -1).astype(np.int32)
        if trial % 3 == 0:                                # brushfire-like: a burst of zeros, then growing priorities
            ops[: n // 4] = 0
        ops = np.concatenate([ops, np.full(n, -1, dtype=np.int32)])
        a = np.zeros(len(ops), dtype=np.uint32)
        b = np.zeros(len(ops), dtype=np.uint32)
        k = L.heap_replay(ops.ctypes.data_as(C.c_void_p), len(ops), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
        assert k > 0 and np.array_equal(a[:k], b[:k]), trial


def test_topdown_pop_leaves_the_array_of_libstdcxx_pop_heap():
    """lds_pop_topdown (helper wave of k_brushfire; scalar statement: heap_pop_topdown in lama_heap.h) stops the descent
    where __push_heap would stop moving the re-inserted entry back up: the resulting ARRAY must equal the one
    std::pop_heap produces after every single operation (ties in later pops depend on the layout)."""
    L = _lib()
    rng = np.random.default_rng(1)
    for trial in range(40):
        n = int(rng.integers(20, 1500))
        span = int(rng.choice([1, 2, 5, 20, 200]))
        ops = np.where(rng.random(n) < 0.55, rng.integers(0, span, size=n), -1).astype(np.int32)
        if trial % 4 == 0:                                # brushfire-like: zeros first, then a slowly growing front
            ops[: n // 5] = 0
            front = np.cumsum(rng.random(n) < 0.02)
            ops = np.where(ops >= 0, np.minimum(ops, 3) + front.astype(np.int32), -1).astype(np.int32)
            ops[: n // 5] = 0
        ops = np.concatenate([ops, np.full(n, -1, dtype=np.int32)])
        assert L.heap_replay_layout(ops.ctypes.data_as(C.c_void_p), len(ops)) == len(ops), trial
