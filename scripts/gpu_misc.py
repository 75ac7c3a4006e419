import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
t = torch.arange(16, dtype=torch.float64, device="cuda")
class Wrap:
    def __init__(self, ptr, n): self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2, "strides": None}
try:
    v = torch.as_tensor(Wrap(t.data_ptr() + 8*4, 8), device="cuda")
    v += 100
    torch.cuda.synchronize()
    print("zero-copy view ok:", t.cpu().numpy(), v.data_ptr() == t.data_ptr() + 32)
except Exception as e:
    print("cuda_array_interface failed:", repr(e))
