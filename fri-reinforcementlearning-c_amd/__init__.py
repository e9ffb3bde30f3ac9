"""frirl_amd -- MI355X-native FRIRL / FIVE hot path (package directory: fri-reinforcementlearning-c_amd/).

The product is the C-ABI shared library lib/libfrirl_hip.so (hand-written gfx950 HIP kernels,
declared in include/frirl_hip.h) plus the ANSI-C drop-in host library that exports the
reference's five_* / FIVE_* / frirl_* API on top of it.  This Python module is only plumbing for
tests and bench.py: a ctypes binding that hands raw device pointers of torch tensors to the C ABI.
There is no CPU fallback: a missing library raises, a missing GPU makes every call return
FRIRL_HIP_ENODEV (raised as FrirlHipError).
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
HIP_LIB_PATH = os.path.join(PKG_DIR, "lib", "libfrirl_hip.so")

NO_HIT = 0xFFFFFFFF
MAX_NANT = 16
MAX_ACTIONS = 32


class FrirlHipError(RuntimeError):
    pass


class Tables(C.Structure):
    """struct frirl_hip_tables (include/frirl_hip.h)."""
    _fields_ = [("nant", C.c_int32), ("U", C.c_int32), ("u", C.c_void_p), ("ve", C.c_void_p)]


class RuleBases(C.Structure):
    """struct frirl_hip_rulebases (include/frirl_hip.h)."""
    _fields_ = [("E", C.c_int32), ("maxR", C.c_int32), ("rb", C.c_void_p), ("nrules", C.c_void_p)]


_lib = None

# name -> (restype, argtypes); every symbol include/frirl_hip.h declares
SIGNATURES = {
    "frirl_hip_version": (C.c_char_p, []),
    "frirl_hip_last_error": (C.c_char_p, []),
    "frirl_hip_device_count": (C.c_int, []),
    "frirl_hip_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "five_hip_rule_distance": (C.c_int, [C.POINTER(Tables), C.POINTER(RuleBases), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def build(force=False, verbose=False):
    import importlib.util
    spec = importlib.util.spec_from_file_location("frirl_amd_build", os.path.join(PKG_DIR, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_all(force=force, verbose=verbose)


def lib():
    """The loaded C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise FrirlHipError(f"{HIP_LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first; "
                                "the FRIRL hot path has no CPU fallback")
        # One HIP runtime per process: torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Loaded
        # first, it satisfies this library's NEEDED entry, so kernels, streams and events all live in the
        # runtime torch uses.  (A plain C host links /opt/rocm's runtime and never sees torch.)
        import torch  # noqa: F401
        L = C.CDLL(HIP_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise FrirlHipError(f"{what}: rc={rc}: {lib().frirl_hip_last_error().decode()}")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


class Problem:
    """Device-resident tables + E rule bases (torch tensors own the HBM; the C ABI gets raw pointers)."""

    def __init__(self, u, ve, rb, nrules):
        import torch
        assert u.is_cuda and ve.is_cuda and rb.is_cuda and nrules.is_cuda
        assert u.dtype == torch.float64 and ve.dtype == torch.float64 and rb.dtype == torch.float64 and nrules.dtype == torch.int32
        self.nant, self.U = u.shape
        self.E, cols, self.maxR = rb.shape
        assert cols == self.nant + 1 and ve.shape == u.shape and nrules.shape == (self.E,)
        self.u, self.ve, self.rb, self.nrules = u.contiguous(), ve.contiguous(), rb.contiguous(), nrules.contiguous()
        self.tables = Tables(self.nant, self.U, self.u.data_ptr(), self.ve.data_ptr())
        self.bases = RuleBases(self.E, self.maxR, self.rb.data_ptr(), self.nrules.data_ptr())

    def rule_distance(self, x, ruledists=None, hit=None, materialise=True, stream=None):
        """five_hip_rule_distance: returns (ruledists [E,maxR] or None, hit [E] int32 with -1 = none)."""
        import torch
        assert x.is_cuda and x.dtype == torch.float64 and x.shape == (self.E, self.nant) and x.is_contiguous()
        if materialise and ruledists is None:
            ruledists = torch.empty((self.E, self.maxR), dtype=torch.float64, device=x.device)
        if hit is None:
            hit = torch.empty((self.E,), dtype=torch.int32, device=x.device)
        rc = lib().five_hip_rule_distance(C.byref(self.tables), C.byref(self.bases), _ptr(x),
                                          _ptr(ruledists) if materialise else None, _ptr(hit), _stream(stream))
        check(rc, "five_hip_rule_distance")
        return (ruledists if materialise else None), hit
