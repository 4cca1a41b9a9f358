"""TEST INFRASTRUCTURE.  A ctypes stand-in for the handful of ``cffi`` calls the reference's Hanabi binding makes
(onpolicy/envs/hanabi/pyhanabi.py: FFI.cdef / dlopen / new / string), so that the reference's own ``pyhanabi`` and
``Hanabi_Env`` modules run unmodified in this container, where cffi is not installed.  oracle/make_golden_hanabi.py
registers it as ``sys.modules['cffi']`` before importing the reference; nothing in the product uses it.

The C header the reference hands to ``cdef`` only declares structs that are either ``{void*}`` handles or
``{int, int}`` cards, and functions over int / bool / pointer arguments -- that is all this understands.
"""
import ctypes
import re


class _Handle(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p)]


class _Card(ctypes.Structure):
    _fields_ = [("color", ctypes.c_int), ("rank", ctypes.c_int)]


class _Pointer(object):
    """What ``ffi.new("T*")`` returns: owns one struct, passes as its address, forwards field access."""

    def __init__(self, obj):
        object.__setattr__(self, "_obj", obj)
        object.__setattr__(self, "_as_parameter_", ctypes.byref(obj))

    def __getattr__(self, name):
        return getattr(self._obj, name)

    def __setattr__(self, name, value):
        setattr(self._obj, name, value)


class _Library(object):
    def __init__(self, path, prototypes):
        self._dll = ctypes.CDLL(path)          # OSError if missing, as cffi's dlopen
        for name, (res, args) in prototypes.items():
            fn = getattr(self._dll, name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)


def _ctype(decl, is_return):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_void_p                 # char* results stay raw addresses: the caller frees them
    if decl.startswith("void"):
        return None
    if decl.startswith("bool"):
        return ctypes.c_bool
    return ctypes.c_int


class FFI(object):
    NULL = None

    def __init__(self):
        self._prototypes = {}

    def cdef(self, text):
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
        for m in re.finditer(r"([\w\s\*]+?)\b(\w+)\s*\(([^)]*)\)\s*;", text):
            res, name, params = m.group(1), m.group(2), m.group(3).strip()
            args = [] if params in ("", "void") else [_ctype(p, False) for p in params.split(",")]
            self._prototypes[name] = (_ctype(res, True), args)

    def dlopen(self, path):
        return _Library(path, self._prototypes)

    def new(self, decl, init=None):
        decl = decl.strip()
        if decl == "char[]":
            return ctypes.create_string_buffer(init)
        m = re.match(r"char\s*\*\s*\[(\d+)\]", decl)
        if m:
            arr = (ctypes.c_char_p * int(m.group(1)))()
            for i, buf in enumerate(init):
                arr[i] = ctypes.cast(buf, ctypes.c_char_p)
            arr._keep = list(init)
            return arr
        if decl == "pyhanabi_card_t*":
            return _Pointer(_Card())
        if re.match(r"pyhanabi_\w+_t\s*\*", decl):
            return _Pointer(_Handle())
        raise NotImplementedError(decl)

    def string(self, address):
        return ctypes.string_at(address)
