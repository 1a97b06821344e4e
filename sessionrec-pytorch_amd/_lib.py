"""ctypes binding of libsrec_hip.so.

The prototypes are parsed from include/srec.h, so every symbol the header declares
is bound (and a missing export fails at import).  There is NO fallback: if the HIP
library is absent, or a tensor is not on the GPU, the product path raises.
"""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'srec.h')
LIB_PATH = os.path.join(HERE, 'libsrec_hip.so')

_CT = {
    'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float,
    'const float*': ctypes.c_void_p, 'float*': ctypes.c_void_p, 'const int*': ctypes.c_void_p,
    'int*': ctypes.c_void_p, 'void*': ctypes.c_void_p, 'const float* const*': ctypes.c_void_p,
    'unsigned char*': ctypes.c_void_p, 'const unsigned char*': ctypes.c_void_p, 'const long long*': ctypes.c_void_p, 'long long*': ctypes.c_void_p, 'const void*': ctypes.c_void_p, 'long*': ctypes.c_void_p, 'const long*': ctypes.c_void_p,
}


def parse_header(path=HEADER):
    """-> {name: [(ctype_string, argname), ...]} for every `int srec_*(...)` declaration."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\bint\s+(srec_\w+)\s*\(([^)]*)\)\s*;', src):
        args = []
        for a in m.group(2).split(','):
            a = ' '.join(a.split())
            mm = re.match(r'^(.*?)(\w+)$', a)
            ty = mm.group(1).strip().replace(' *', '*')
            args.append((ty, mm.group(2)))
        protos[m.group(1)] = args
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libsrec_hip.so is missing (%s): build it with `python -c "import __graft_entry__ as g; g.build()"`. '
                'There is no CPU fallback for the product path.' % LIB_PATH)
        dll = ctypes.CDLL(LIB_PATH)
        for name, args in self.protos.items():
            fn = getattr(dll, name)            # AttributeError if the export is missing
            fn.restype = ctypes.c_int
            fn.argtypes = [_CT[t] for t, _ in args]
        self._dll = dll
        return dll

    def __getattr__(self, name):
        if name.startswith('srec_'):
            fn = getattr(self.load(), name)

            def call(*a):
                rc = fn(*a)
                if rc != 0:
                    raise RuntimeError('%s failed with status %d%s' % (
                        name, rc, ' (bad argument: alignment / dimension contract of include/srec.h)' if rc == 1001 else
                        ' (hipError_t)'))
            return call
        raise AttributeError(name)


lib = _Lib()


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """device pointer of a tensor (None -> NULL).  Raises for CPU tensors: no fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('sessionrec-pytorch_amd ops need GPU (HIP) tensors; got a %s tensor' % t.device)
    return t.data_ptr()


def f32c(t):
    assert t.dtype == torch.float32, t.dtype
    return t if t.is_contiguous() else t.contiguous()
