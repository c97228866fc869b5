"""CPU tests of the drop-in boundary: liblbm_hip.so loads without a GPU, exports every entry
point include/lbm.h declares, the ctypes struct matches the C layout, and the product path
fails loudly (no CPU fallback) when there is no device.  No compute call is made here."""
import ctypes
import os
import re

import pytest

from latticeboltzmannsimulations_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "lbm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lbm_[a-z_0-9]+)\s*\(", src)))


def test_header_and_bindings_list_the_same_entry_points():
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(_lib.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "liblbm_hip.so not built (run __graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in declared_functions():
        assert hasattr(L, n), f"{n} declared in include/lbm.h but not exported"
    hdr = int(re.search(r"#define LBM_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "lbm.h")).read()).group(1))
    assert _lib.lib().lbm_abi_version() == hdr == _lib.ABI_VERSION


def test_library_has_no_static_rccl_dependency():
    """RCCL is bound lazily (dlopen) so that single-GPU processes never load it and so that
    torch.distributed and this library share ONE RCCL / HIP runtime in a process."""
    import subprocess
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in out
    assert "libamdhip64" in out


def test_params_struct_layout():
    assert ctypes.sizeof(_lib.lbm_params) == 18 * 4 + 6 * 8
    assert _lib.lbm_params.batch.offset == 48
    assert _lib.lbm_params.ny_local_min.offset == 56 and _lib.lbm_params.flags.offset == 68
    assert _lib.lbm_params.uLB.offset == 72


def test_enums_match_header():
    src = open(os.path.join(ROOT, "include", "lbm.h")).read()
    for name in ("LBM_F32", "LBM_F64", "LBM_SRT", "LBM_TRT", "LBM_MRT", "LBM_SEM_MRT_PY", "LBM_SEM_MRT_GPU",
                 "LBM_KERNEL_AUTO", "LBM_KERNEL_GENERIC", "LBM_KERNEL_VEC", "LBM_KERNEL_TB", "LBM_KERNEL_PUSH", "LBM_KERNEL_STREAM", "LBM_SIDE_LOW", "LBM_SIDE_HIGH",
                 "LBM_LAYOUT_AUTO", "LBM_LAYOUT_PLANES", "LBM_LAYOUT_ROWS", "LBM_ARITH_STRICT", "LBM_ARITH_FAST",
                 "LBM_FLAG_NO_DEEP_HALO", "LBM_FLAG_FRAME_UNFUSED", "LBM_FLAG_FRAME_FUSED_BATCH", "LBM_FLAG_NO_FRAME_LDS",
                 "LBM_FLAG_NT_ON", "LBM_FLAG_NT_OFF", "LBM_FLAG_COMM_PRIORITY_OFF", "LBM_FLAG_EAGER_LAG", "LBM_FLAG_FRAME_BESIDE_ON",
                 "LBM_FLAG_FRAME_BESIDE_OFF", "LBM_FLAG_FRAME_NARROW", "LBM_FLAG_NO_EDGE_FIRST", "LBM_FLAG_NO_EDGE_RESERVE", "LBM_FLAG_NO_XCD_BANDS", "LBM_FLAG_NO_TAIL_TILES"):
        m = re.search(name + r"\s*=\s*(-?\d+)", src)
        assert m and int(m.group(1)) == getattr(_lib, name), name


def test_null_arguments_are_rejected_not_dereferenced():
    L = _lib.lib()
    assert L.lbm_comm_unique_id(None) == -1            # LBM_ERR_INVALID
    assert L.lbm_step(None, 1) == -1
    assert L.lbm_set_relaxation(None, 0, 1.0, 1.0, 1.0, 1.0, 1.0) == -1
    assert L.lbm_sync(None) == -1
    assert L.lbm_steps_done(None) == -1
    assert L.lbm_halo_elems(None) == 0
    assert L.lbm_last_error(None) == b"null context"
    L.lbm_destroy(None)


def test_create_fails_loudly_without_a_device():
    L = _lib.lib()
    if L.lbm_device_count() > 0:
        pytest.skip("a GPU is present")
    from latticeboltzmannsimulations_amd import CavitySolver
    with pytest.raises(RuntimeError, match="no HIP device"):
        CavitySolver(64, 64, 100.0)


def test_create_rejects_bad_parameters():
    L = _lib.lib()
    err = ctypes.create_string_buffer(256)
    p = _lib.lbm_params()
    p.struct_size = 3
    assert not L.lbm_create(ctypes.byref(p), err, len(err))
    assert b"struct_size" in err.value
    p.struct_size = ctypes.sizeof(_lib.lbm_params)
    p.nx, p.ny, p.y0, p.ny_local = 64, 64, 0, 64
    p.dtype, p.collision, p.semantics, p.turb = 1, 2, 0, 1     # Smagorinsky exists only with MRT_GPU semantics
    assert not L.lbm_create(ctypes.byref(p), err, len(err))
    assert b"turb" in err.value
    p.turb, p.semantics, p.ny_local = 0, 1, 1
    assert not L.lbm_create(ctypes.byref(p), err, len(err))
    assert b"slab" in err.value


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: no product file (Python or HIP) imports, includes, links or opens anything under
    oracle/ -- checked on the import / include / path forms, not on prose."""
    pkg = os.path.join(ROOT, "latticeboltzmannsimulations_amd")
    bad = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+\.+\s*oracle\b|#\s*include\s*[\"<][^\">]*oracle)|liblbmref|lbm_ref\b|lbm_numpy\b|"
                     r"[\"\'][^\"\']*oracle/[^\"\']*[\"\']", re.M)
    seen = 0
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                seen += 1
                m = bad.search(open(os.path.join(dirpath, f)).read())
                assert m is None, (f, m.group(0))
    assert seen >= 10


def test_library_reads_no_tuning_from_the_environment():
    """A/B switches travel in lbm_params (tb_steps, frame_seg, flags), not in ambient state a caller cannot see in the ABI:
    the only getenv left in the HIP sources is behind #ifdef LBM_DEBUG (a timing diagnostic of debug builds)."""
    csrc = os.path.join(ROOT, "latticeboltzmannsimulations_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".hpp")):
            continue
        lines = open(os.path.join(csrc, f)).read().split("\n")
        depth_debug = 0
        for ln in lines:
            if ln.startswith("#ifdef LBM_DEBUG"):
                depth_debug += 1
            elif ln.startswith("#else") or ln.startswith("#endif"):
                depth_debug = max(0, depth_debug - 1)
            elif "getenv" in ln:
                assert depth_debug > 0, (f, ln.strip())


def test_integration_md_binding_matches_the_struct():
    """The ctypes stub shown in INTEGRATION.md must describe the same struct as include/lbm.h / _lib.lbm_params."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"c_int32\) for n in \((.*?)\)\] \+.*?c_double\) for n in \((.*?)\)\]", txt, re.S)
    ints = re.findall(r'"(\w+)"', m.group(1))
    dbls = re.findall(r'"(\w+)"', m.group(2))
    assert ints + dbls == [f[0] for f in _lib.lbm_params._fields_]
    hdr = open(os.path.join(ROOT, "include", "lbm.h")).read()
    body = hdr[hdr.index("typedef struct lbm_params {"):hdr.index("} lbm_params;")]
    assert re.findall(r"^\s+(?:int32_t|double)\s+(\w+);", body, re.M) == ints + dbls


def test_documented_limits_follow_the_source_constants():
    """The limits that include/lbm.h and the error messages quote (rows of a deep halo, steps per launch) are derived from
    GHY (lbm_device.hpp) and ST_MAX_S (lbm_stream.hpp), not typed in a second time ('1 <= nrows <= 5' had gone stale)."""
    import re
    csrc = os.path.join(ROOT, "latticeboltzmannsimulations_amd", "csrc")
    dev = open(os.path.join(csrc, "lbm_device.hpp")).read()
    st = open(os.path.join(csrc, "lbm_stream.hpp")).read()
    ghy = int(re.search(r"constexpr int GHY = (\d+);", dev).group(1))
    waves = int(re.search(r"constexpr int ST_WAVES = (\d+);", st).group(1))
    assert re.search(r"constexpr int ST_MAX_S = ST_WAVES / 2;", st)
    max_s = waves // 2
    hdr = open(os.path.join(ROOT, "include", "lbm.h")).read()
    assert f"1 <= nrows <= {ghy - 1}" in hdr and f"carries {ghy} ghost rows" in hdr
    max_p = int(re.search(r"constexpr int SP_MAX_WAVES = (\d+);", st).group(1)) - 2
    assert re.search(r"constexpr int SP_MAX_S = SP_MAX_WAVES - 2;", st)
    assert f"(at most {max_s};" in hdr and f"STREAM: 2 .. {max_s}; a lone MRT_GPU lattice, two rows per wave: 2 .. {max_p}" in hdr
    src = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith(".hip"))
    assert "std::to_string(GHY - 1)" in src and "std::to_string(ST_MAX_S)" in src and "std::to_string(SP_MAX_S)" in src
    assert not re.search(r"<= nrows <= \d", src), "a literal row limit in an error message"
