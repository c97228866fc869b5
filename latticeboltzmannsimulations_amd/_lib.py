"""Loader for liblbm_hip.so (the C ABI of include/lbm.h) -- ctypes only, no torch types.

The product path has NO CPU fallback: if the shared library is missing or cannot be
loaded, every entry point raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("LBM_LIB_PATH", os.path.join(_HERE, "liblbm_hip.so"))   # override: A/B builds of the kernels
HEADER = os.path.join(os.path.dirname(_HERE), "include", "lbm.h")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall"]
LINK_FLAGS = ["-ldl"]   # RCCL is bound lazily with dlopen inside the library

LBM_F32, LBM_F64 = 0, 1
LBM_SRT, LBM_TRT, LBM_MRT = 0, 1, 2
LBM_SEM_MRT_PY, LBM_SEM_MRT_GPU = 0, 1
LBM_KERNEL_AUTO, LBM_KERNEL_GENERIC, LBM_KERNEL_VEC, LBM_KERNEL_TB, LBM_KERNEL_PUSH, LBM_KERNEL_STREAM = 0, 1, 2, 3, 4, 5
LBM_LAYOUT_AUTO, LBM_LAYOUT_PLANES, LBM_LAYOUT_ROWS = 0, 1, 2
LBM_SIDE_LOW, LBM_SIDE_HIGH = 0, 1
LBM_ARITH_STRICT, LBM_ARITH_FAST = 0, 1
# lbm_params.flags (A/B switches of the launch plan; results never depend on them)
LBM_FLAG_NO_DEEP_HALO, LBM_FLAG_FRAME_UNFUSED, LBM_FLAG_FRAME_FUSED_BATCH, LBM_FLAG_NO_FRAME_LDS = 1, 2, 4, 8
LBM_FLAG_NT_ON, LBM_FLAG_NT_OFF, LBM_FLAG_COMM_PRIORITY_OFF, LBM_FLAG_EAGER_LAG = 16, 32, 64, 128
LBM_FLAG_FRAME_BESIDE_ON, LBM_FLAG_FRAME_BESIDE_OFF, LBM_FLAG_FRAME_NARROW, LBM_FLAG_NO_EDGE_FIRST = 512, 1024, 2048, 4096
LBM_FLAG_NO_EDGE_RESERVE, LBM_FLAG_NO_XCD_BANDS, LBM_FLAG_NO_TAIL_TILES, LBM_FLAG_STREAM_WALLS = 8192, 16384, 32768, 65536
LBM_FLAG_STREAM_PAIRS, LBM_FLAG_NO_STREAM_WALLS = 131072, 262144
ABI_VERSION = 3        # = LBM_ABI_VERSION of include/lbm.h (tests/test_abi.py keeps them equal)


class lbm_params(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("nx", ctypes.c_int32), ("ny", ctypes.c_int32),
                ("y0", ctypes.c_int32), ("ny_local", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("collision", ctypes.c_int32), ("semantics", ctypes.c_int32), ("kernel", ctypes.c_int32),
                ("turb", ctypes.c_int32), ("device", ctypes.c_int32), ("layout", ctypes.c_int32),
                ("batch", ctypes.c_int32), ("arith", ctypes.c_int32), ("ny_local_min", ctypes.c_int32),
                ("tb_steps", ctypes.c_int32), ("frame_seg", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("uLB", ctypes.c_double), ("omega", ctypes.c_double), ("omegam", ctypes.c_double),
                ("omega_e", ctypes.c_double), ("omega_eps", ctypes.c_double), ("omega_q", ctypes.c_double)]


def sources():
    return [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".hip", ".hpp"))] + [HEADER]


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into liblbm_hip.so next to this file (in-tree).  The translation units (four of host code +
    C ABI: lbm_hip / lbm_plan / lbm_launch / lbm_comm; the explicit instantiations of the tile and streaming kernels for float and
    for double) are compiled in parallel into csrc/_obj/ and linked."""
    srcs = sources()
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(_CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    units = [s for s in srcs if s.endswith(".hip")]
    objs = [os.path.join(objdir, os.path.basename(u)[:-4] + ".o") for u in units]

    def compile_one(pair):
        cmd = [HIPCC] + cflags + ["-c", pair[0], "-o", pair[1]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 4)) as pool:
        list(pool.map(compile_one, zip(units, objs)))
    cmd = [HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH] + objs + LINK_FLAGS
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None

# name -> (restype, argtypes); exactly the entry points declared in include/lbm.h
_vp, _i, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
SIGNATURES = {
    "lbm_abi_version": (_i, []),
    "lbm_device_count": (_i, []),
    "lbm_create": (_vp, [ctypes.POINTER(lbm_params), ctypes.c_char_p, ctypes.c_size_t]),
    "lbm_destroy": (None, [_vp]),
    "lbm_last_error": (ctypes.c_char_p, [_vp]),
    "lbm_init_equilibrium": (_i, [_vp]),
    "lbm_set_state": (_i, [_vp, _vp, _i]),
    "lbm_set_relaxation": (_i, [_vp, _i, _d, _d, _d, _d, _d]),
    "lbm_step": (_i, [_vp, _i]),
    "lbm_sync": (_i, [_vp]),
    "lbm_time_steps": (_i, [_vp, _i, ctypes.POINTER(_d)]),
    "lbm_steps_done": (ctypes.c_longlong, [_vp]),
    "lbm_next_unit": (_i, [_vp, _i]),
    "lbm_describe": (_i, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
    "lbm_plan": (_i, [ctypes.POINTER(lbm_params), _i, _i, ctypes.c_char_p, ctypes.c_size_t]),
    "lbm_get_fields": (_i, [_vp, _vp, _vp, _vp, _i]),
    "lbm_mean_u": (_i, [_vp, ctypes.POINTER(_d)]),
    "lbm_get_tau": (_i, [_vp, _vp, _i]),
    "lbm_halo_elems": (_i, [_vp]),
    "lbm_halo_export": (_i, [_vp, _i, _vp]),
    "lbm_halo_import": (_i, [_vp, _i, _vp]),
    "lbm_step_edges": (_i, [_vp]),
    "lbm_step_interior": (_i, [_vp]),
    "lbm_step_finish": (_i, [_vp]),
    "lbm_halo_rows_elems": (ctypes.c_longlong, [_vp, _i]),
    "lbm_halo_export_rows": (_i, [_vp, _i, _i, _vp]),
    "lbm_halo_import_rows": (_i, [_vp, _i, _i, _vp]),
    "lbm_step_unit": (_i, [_vp, _i]),
    "lbm_comm_unique_id": (_i, [_vp]),
    "lbm_comm_init": (_i, [_vp, _i, _i, _vp]),
    "lbm_comm_loopback": (_i, [_vp]),
    "lbm_copy_bandwidth": (_i, [_vp, ctypes.c_size_t, _i, ctypes.POINTER(_d)]),
    "lbm_fma_rate": (_i, [_vp, _d, ctypes.POINTER(_d)]),
}


def _preload_torch_runtime(names=("libamdhip64.so",)):
    """One process can hold only ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64 / librccl (torch/lib); if liblbm_hip.so pulled in /opt/rocm's copies first, a
    later `import torch` would find no GPU (measured on the MI355X box).  So when torch is
    installed, load ITS runtime libraries by path first (without importing torch); the
    DT_NEEDED entries of liblbm_hip.so then resolve to them by SONAME, and torch.distributed
    (RCCL) and this library share one runtime whichever is imported first."""
    if os.environ.get("LBM_USE_SYSTEM_ROCM"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in names:
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def lib():
    """The loaded library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _preload_torch_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if an exported symbol is missing
            f.restype, f.argtypes = res, args
        if L.lbm_abi_version() != ABI_VERSION:
            raise RuntimeError("liblbm_hip.so ABI version mismatch")
        _lib = L
    return _lib
