"""ctypes binding of the C ABI declared in include/r8bsrc.h.

`bind(path)` loads a shared library exporting that ABI and sets the prototypes.  The package
itself only ever binds r8brain-free-src_amd/libr8bsrc_hip.so (see `load()`): if that library or a
HIP device is missing the package raises -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libr8bsrc_hip.so"

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)

# every symbol include/r8bsrc.h declares: (name, restype, argtypes)
PROTOTYPES = [
    ("r8b_create", C.c_void_p, [C.c_double, C.c_double, C.c_int, C.c_double, C.c_int]),
    ("r8b_delete", None, [C.c_void_p]),
    ("r8b_inlen", C.c_int, [C.c_void_p, C.c_int]),
    ("r8b_clear", None, [C.c_void_p]),
    ("r8b_process", C.c_int, [C.c_void_p, dp, C.c_int, C.POINTER(dp)]),
    ("r8b_batch_create", C.c_void_p, [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                      C.c_int, C.c_int]),
    ("r8b_batch_create_ex", C.c_void_p, [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                         C.c_int, C.c_int, C.c_int]),
    ("r8b_batch_delete", None, [C.c_void_p]),
    ("r8b_batch_clear", None, [C.c_void_p]),
    ("r8b_batch_channels", C.c_int, [C.c_void_p]),
    ("r8b_batch_device", C.c_int, [C.c_void_p]),
    ("r8b_batch_max_out_len", C.c_int, [C.c_void_p]),
    ("r8b_batch_inlen", C.c_int, [C.c_void_p, C.c_int]),
    ("r8b_batch_inlen_before_outpos", C.c_int, [C.c_void_p, C.c_int]),
    ("r8b_batch_process", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p,
                                    C.c_longlong, C.c_void_p]),
    ("r8b_batch_process_host", C.c_int, [C.c_void_p, dp, C.c_longlong, C.c_int, dp,
                                         C.c_longlong]),
    ("r8b_batch_process_pcm", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong,
                                        C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_longlong,
                                        C.c_void_p]),
    ("r8b_pcm_sample_bytes", C.c_int, [C.c_int]),
    ("r8b_batch_state_size", C.c_longlong, [C.c_void_p]),
    ("r8b_batch_state_save", C.c_longlong, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    ("r8b_batch_state_load", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    ("r8b_batch_create_stage", C.c_void_p, [C.c_int, C.c_double, C.c_double, C.c_double,
                                            C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int]),
    ("r8b_batch_describe", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("r8b_batch_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("r8b_batch_stage_count", C.c_int, [C.c_void_p]),
    ("r8b_batch_stat", C.c_longlong, [C.c_void_p, C.c_char_p]),
    ("r8b_batch_latency_frac", C.c_double, [C.c_void_p]),
    ("r8b_batch_stage_timing", C.c_int, [C.c_void_p, C.c_int, dp, ip, C.POINTER(C.c_longlong),
                                C.POINTER(C.c_longlong), C.c_char_p, C.c_int]),
    ("r8b_batch_stage_symbol", C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    ("r8b_last_error", C.c_char_p, []),
    ("r8b_design_lpfilter", C.c_int, [C.c_double, C.c_double, C.c_double, C.c_double, ip, ip, dp,
                                      C.c_int]),
    ("r8b_design_lpfilter_ex", C.c_int, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.c_int]),
    ("r8b_design_fracbank", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, ip, ip, dp,
                                      C.c_int]),
    ("r8b_design_hbfilter", C.c_int, [C.c_double, C.c_int, C.c_int, dp, dp]),
    ("r8b_design_whole_stepping", C.c_int, [C.c_double, C.c_double, ip, ip]),
    ("r8b_design_cache_counts", None, [ip, ip, ip]),
    ("r8b_plan_create", C.c_void_p, [C.c_double, C.c_double, C.c_int, C.c_double, C.c_double]),
    ("r8b_plan_delete", None, [C.c_void_p]),
    ("r8b_plan_clear", None, [C.c_void_p]),
    ("r8b_plan_step", C.c_int, [C.c_void_p, C.c_int]),
    ("r8b_plan_max_out_len", C.c_int, [C.c_void_p]),
    ("r8b_plan_inlen", C.c_int, [C.c_void_p, C.c_int]),
    ("r8b_plan_inlen_before_outpos", C.c_int, [C.c_void_p, C.c_int]),
    ("r8b_plan_describe", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("r8b_version", C.c_char_p, []),
]


# parity-test hook (include/r8bsrc.h r8b_lp_provider)
LP_PROVIDER = C.CFUNCTYPE(C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double),
                          C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int))


# ... which only TEST builds of the library export (-DR8B_TEST_HOOKS: tests/emul, tests/_build)
TEST_HOOK_PROTOTYPES = [
    ("r8b_design_set_lp_provider", None, [C.c_void_p]),
]


def bind(path, test_hooks=False):
    lib = C.CDLL(path)
    for name, res, args in PROTOTYPES + (TEST_HOOK_PROTOTYPES if test_hooks else []):
        f = getattr(lib, name)  # AttributeError if the library does not export it
        f.restype = res
        f.argtypes = args
    return lib


_lib = None


def lib_path():
    # R8B_HIP_LIB: kernel-tuning experiments load an alternative BUILD of the same HIP library
    return os.environ.get("R8B_HIP_LIB") or os.path.join(_HERE, LIB_NAME)


def load():
    """The product library.  Raises if it was not built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError("%s not found: build it with `make -C %s` (or "
                               "__graft_entry__.build()); there is no CPU fallback" %
                               (p, os.path.join(_HERE, "csrc")))
        # A process that also uses PyTorch-ROCm must let torch load ITS HIP runtime first: the torch
        # wheel bundles its own libamdhip64, and if this library pulls in the system one before, torch
        # finds no device afterwards (measured on the GPU box: torch.cuda.is_available() turns False).
        # Loaded after torch, libr8bsrc_hip.so resolves against the runtime already in the process.
        # Hosts without torch are unaffected; C / C++ hosts link the system runtime as usual.
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _lib = bind(p)
    return _lib
