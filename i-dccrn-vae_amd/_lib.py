"""ctypes binding of libidccrn_hip.so (the C ABI declared in include/idccrn_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no
CPU fallback: if the shared object is missing or a symbol is absent the import of the
operators fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# IDV_LIB_PATH: another build of the same library (A/B measurements of a kernel variant on one box); default: the in-tree build
LIB_PATH = os.environ.get("IDV_LIB_PATH") or os.path.join(_HERE, "csrc", "libidccrn_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "idccrn_hip.h")

_lib = None


class IdvError(RuntimeError):
    pass


def declared_symbols() -> list:
    """Every function name declared in include/idccrn_hip.h."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(idv_[a-z0-9_]+)\s*\(", src)))


_CTYPES = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "float": ctypes.c_float, "double": ctypes.c_double}


def prototypes() -> dict:
    """name -> (return C type, [parameter C types]) of every function declared in include/idccrn_hip.h; a pointer of any
    kind is the string "ptr"."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//.*", "", src)
    out = {}
    for ret, name, args in re.findall(r"\b(int|long long|void)\s+(idv_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        params = []
        for a in args.split(","):
            a = a.strip()
            if not a or a == "void":
                continue
            t = re.sub(r"\b[a-zA-Z_][a-zA-Z0-9_]*$", "", a).strip()
            if t.endswith("*"):
                params.append("ptr")
            else:
                t = t.replace("const ", "").strip()
                if t not in _CTYPES:
                    raise IdvError(f"include/idccrn_hip.h: unknown parameter type {t!r} in {name}")
                params.append(t)
        out[name] = (ret, params)
    return out


def declared_abi_version() -> int:
    with open(HEADER_PATH) as f:
        m = re.search(r"#define\s+IDV_ABI_VERSION\s+(\d+)", f.read())
    if not m:
        raise IdvError("include/idccrn_hip.h does not define IDV_ABI_VERSION")
    return int(m.group(1))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IdvError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP hot path.")
        _lib = ctypes.CDLL(LIB_PATH)
        want = declared_abi_version()
        if not hasattr(_lib, "idv_abi_version") or int(_lib.idv_abi_version()) != want:
            got = int(_lib.idv_abi_version()) if hasattr(_lib, "idv_abi_version") else None
            _lib = None
            raise IdvError(f"{LIB_PATH} has ABI version {got}, include/idccrn_hip.h declares {want}: rebuild it "
                           "(`python __graft_entry__.py`)")
        protos = prototypes()
        for name in declared_symbols():
            if not hasattr(_lib, name):
                if os.environ.get("IDV_DEV_PARTIAL_LIB"):
                    continue
                raise IdvError(f"libidccrn_hip.so does not export {name}")
            if name in protos:
                # the header's prototypes become ctypes signatures: a `long long` / `double` argument can then not be
                # truncated by a call site that forgot its ll() / d() wrapper, and a wrong argument count raises
                ret, params = protos[name]
                fn = getattr(_lib, name)
                fn.argtypes = [ctypes.c_void_p if t == "ptr" else _CTYPES[t] for t in params]
                fn.restype = None if ret == "void" else _CTYPES[ret]
    return _lib


_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_longlong
_F = ctypes.c_float
_D = ctypes.c_double


def call(name: str, *args):
    """Call an idv_* entry; tensors become device pointers, python numbers keep their C type
    via the wrappers p()/i()/f()/d()/ll().  Raises IdvError on a non-zero status."""
    fn = getattr(lib(), name)
    # wrapped scalars go in by VALUE: the prototype (argtypes) decides the C type, so i(n) for a `long long` parameter widens
    rc = fn(*[a.value if isinstance(a, (_I, _L, _F, _D)) else a for a in args])
    if rc != 0:
        what = {-1: "invalid argument", -2: "launch failure",
                -3: "a cooperative recurrence of this device timed out earlier (a sibling workgroup never became resident); its "
                    "outputs are NaN-poisoned, this call was not launched and none will be until ops.coop_clear() / "
                    "idv_coop_last_status(1) acknowledges the fault"}.get(rc, "?")
        raise IdvError(f"{name} failed with status {rc} ({what})")


class _TensorPtr(ctypes.c_void_p):
    """c_void_p that keeps its tensor alive until the (asynchronous, stream-ordered) call is issued."""


def p(t) -> ctypes.c_void_p:
    """Device pointer of a tensor (None -> NULL).  The returned object holds a reference to the
    tensor, so temporaries (``x.contiguous()``, ``x.to(dev)``) survive until the launch is queued;
    after that the caching allocator's stream ordering protects the memory."""
    if t is None:
        return _P(None)
    r = _TensorPtr(t.data_ptr())
    r._keep = t
    return r


def i(v) -> ctypes.c_int:
    return _I(int(v))


def ll(v) -> ctypes.c_longlong:
    return _L(int(v))


def f(v) -> ctypes.c_float:
    return _F(float(v))


def d(v) -> ctypes.c_double:
    return _D(float(v))


def stream_ptr() -> ctypes.c_void_p:
    import torch
    return _P(torch.cuda.current_stream().cuda_stream)
