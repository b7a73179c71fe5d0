"""Device context: one HIP stream + a reusable HBM workspace (treee_ctx in the C ABI).

PyTorch is plumbing only: it owns device tensors and the stream the kernels are enqueued on.
"""
import ctypes as C

import numpy as np

from . import _lib

ENTRY_DTYPE = np.dtype([("distance", "<f4"), ("_pad", "<u4"), ("index", "<u8")])
NODE_DTYPE = np.dtype(
    [(n, "<u4") for n in ("key_prefix", "level", "first", "count", "first_child", "child_mask", "parent", "lod_first")]
)
INVALID_INDEX = np.uint64(0xFFFFFFFFFFFFFFFF)
F32_MAX = float(np.finfo(np.float32).max)


def ptr(a):
    """raw pointer of a numpy array or a torch tensor (None -> NULL)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())


def is_device(a):
    return not isinstance(a, np.ndarray) and hasattr(a, "is_cuda") and a.is_cuda


def host_f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(-1, cols) if cols is not None else a


def dev_f32(t, cols=None):
    import torch

    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t.reshape(-1, cols) if cols is not None else t


class _DeviceMemory:
    def __init__(self, address, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": typestr, "data": (int(address), False), "version": 2}


def device_view(address, shape, dtype, device=None):
    """a torch tensor over device memory the library owns (treee_octree_dev_ptr and friends): no copy; valid as long as the
    handle is, and the caller orders its use against the handle's stream (include/treee_hip.h, "Streams")"""
    import torch

    typestr = {"int32": "<i4", "uint32": "<u4", "float32": "<f4", "int64": "<i8", "uint8": "|u1"}[dtype]
    return torch.as_tensor(_DeviceMemory(address, shape, typestr), device=device or "cuda")


class Context:
    """treee_ctx bound to a torch device and (by default) torch's current stream."""

    def __init__(self, device=None, stream=None):
        import torch

        if not torch.cuda.is_available():
            raise _lib.TreeeError(_lib.TREEE_ERR_HIP, "no HIP device visible to torch; treee_amd has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        elif not isinstance(device, int):  # a torch.device or "cuda:N"
            idx = torch.device(device).index
            device = torch.cuda.current_device() if idx is None else idx
        self.device = torch.device("cuda", device)
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream() if stream is None else stream
        self.stream = s
        h = C.c_void_p()
        _lib.check(_lib.lib().treee_ctx_create(self.device.index, C.c_void_p(s.cuda_stream), C.byref(h)))
        self._h = h

    @property
    def handle(self):
        return self._h

    def sync(self):
        _lib.check(_lib.lib().treee_ctx_sync(self._h))

    def reserve(self, nbytes):
        _lib.check(_lib.lib().treee_ctx_reserve(self._h, int(nbytes)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().treee_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def default_context(device=None):
    import torch

    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    stream = torch.cuda.current_stream(idx)
    key = (idx, stream.cuda_stream)
    if key not in _default:
        _default[key] = Context(idx, stream)
    return _default[key]
