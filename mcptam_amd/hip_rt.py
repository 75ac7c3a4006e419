"""The three HIP runtime calls a Python caller needs to keep an image ring in device memory (a native caller uses hipMalloc /
hipMemcpy directly).  Binds the runtime libmcptam_hip.so itself is linked against, so no second runtime enters the process."""
import ctypes

_rt = None


def _lib():
    global _rt
    if _rt is None:
        from . import chain_bundle
        chain_bundle.lib()                      # libmcptam_hip.so pulls the runtime in
        for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _rt = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _rt is None:
            raise RuntimeError("HIP runtime library not found")
        _rt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _rt.hipFree.argtypes = [ctypes.c_void_p]
        _rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _rt.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        _rt.hipHostFree.argtypes = [ctypes.c_void_p]
    return _rt


class _Pinned:
    """owner of one hipHostMalloc block (freed with the last array that views it)"""
    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        rc = _lib().hipHostMalloc(ctypes.byref(p), max(int(nbytes), 1), 0)
        if rc != 0:
            raise RuntimeError("hipHostMalloc failed (%d)" % rc)
        self.ptr, self.nbytes = p.value, max(int(nbytes), 1)
        self.buf = (ctypes.c_uint8 * self.nbytes).from_address(self.ptr)

    def __del__(self):
        try:
            _lib().hipHostFree(ctypes.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_zeros(count, dtype):
    """numpy array of `count` zero elements in pinned (page-locked) host memory: a result buffer the library's device-to-host copies
    reach by DMA instead of through the runtime's staging of pageable memory (what a native caller gets from hipHostMalloc /
    hipHostRegister on its own arrays)."""
    import numpy as np
    dt = np.dtype(dtype)
    own = _Pinned(int(count)*dt.itemsize)
    a = np.frombuffer(own.buf, dtype=dt, count=int(count))
    ctypes.memset(own.ptr, 0, own.nbytes)
    return a, own


def dev_alloc(nbytes):
    p = ctypes.c_void_p()
    rc = _lib().hipMalloc(ctypes.byref(p), int(nbytes))
    if rc != 0:
        raise RuntimeError("hipMalloc failed (%d)" % rc)
    return p.value


def dev_upload(ptr, array):
    rc = _lib().hipMemcpy(ctypes.c_void_p(ptr), array.ctypes.data, array.nbytes, 1)      # hipMemcpyHostToDevice
    if rc != 0:
        raise RuntimeError("hipMemcpy failed (%d)" % rc)


def dev_free(ptr):
    _lib().hipFree(ctypes.c_void_p(ptr))
