#!/usr/bin/env python3
"""Host cost of the per-launch helpers of the Python binding (gencomm_amd/runtime.py), microseconds per call on this box:
stream handle through a torch.cuda.Stream object against torch's raw accessor, data_ptr, one small ctypes call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gencomm_amd import _lib
from gencomm_amd.runtime import ptr, stream_ptr

dev = torch.device("cuda:0")
x = torch.zeros(16, device=dev)
N = 200000


def t(f):
    f()
    t0 = time.perf_counter()
    for _ in range(N):
        f()
    return (time.perf_counter() - t0) / N * 1e6


print(f"torch.cuda.current_stream(dev).cuda_stream : {t(lambda: torch.cuda.current_stream(dev).cuda_stream):.2f} us")
print(f"runtime.stream_ptr(dev)                     : {t(lambda: stream_ptr(dev)):.2f} us")
print(f"runtime.ptr(tensor)                         : {t(lambda: ptr(x)):.2f} us")
l = _lib.lib()
print(f"ctypes gencomm_abi_version()                : {t(lambda: l.gencomm_abi_version()):.2f} us")
print(f"ctypes gencomm_get_mode(0)                  : {t(lambda: l.gencomm_get_mode(0)):.2f} us")
print(f"torch.empty(64, device)                     : {t(lambda: torch.empty(64, device=dev)):.2f} us")
assert stream_ptr(dev) == torch.cuda.current_stream(dev).cuda_stream
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    assert stream_ptr(dev) == s.cuda_stream
print("raw accessor agrees with the Stream object on the default and on a side stream")
