"""Test infrastructure: the test-suite's OWN build of the host library (the product sources of iris_lama_amd/host compiled
with -DLAMA_TESTING by tests/cpu_engine/Makefile).  Only that build has `lama_host_set_engine_library`, the hook that binds
another implementation of the device C-ABI -- the oracle-backed test double tests/cpu_engine -- so that the host / multi-rank
logic runs on machines without a GPU.  The shipped iris_lama_amd/lib/liblama_host.so has no such hook (tests/test_cabi.py)."""
import ctypes as C
import os
import subprocess

import iris_lama_amd.ffi as F

HERE = os.path.dirname(os.path.abspath(__file__))
CPU_ENGINE = os.path.join(HERE, "cpu_engine", "_build", "liblama_cpu_engine.so")
TEST_HOST = os.path.join(HERE, "cpu_engine", "_build", "liblama_host_testing.so")


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "cpu_engine")], check=True)


def set_engine_library(path):
    """Switch ffi to the testing host library and bind `path` as the device C-ABI; None: unbind and go back to the product library."""
    if path is None:
        if F.HOST_LIB == TEST_HOST:
            F._hostlib().lama_host_set_engine_library(None)
        F.use_host_library(None)
        return
    build()
    F.use_host_library(TEST_HOST)
    L = F._hostlib()
    L.lama_host_set_engine_library.restype = C.c_int32
    L.lama_host_set_engine_library.argtypes = [C.c_char_p]
    if L.lama_host_set_engine_library(path.encode()) != 0:
        raise F.LamaError(f"cannot bind engine library {path}")
