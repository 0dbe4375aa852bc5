"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports
every symbol include/pda_pointnet2.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pdanet_amd import build, _lib
    build.build()
    return _lib.load()


def _declared_symbols():
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("pda_pointnet2.h", "pda_train.h", "pda_pointnet2_stack.h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pda_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), "libpda_pointnet2.so does not export %s" % n


def test_binding_covers_header():
    from pdanet_amd import _lib
    assert sorted(list(_lib.SIGNATURES) + _lib.INFO_SYMBOLS) == _declared_symbols()


def test_info_entry_points(lib, oracle):
    assert lib.pda_abi_version() == 20
    assert lib.pda_fp_contract_mode() == 1
    for n in [1, 2, 3, 7, 8, 100, 1000, 1023, 1024, 4096, 16384, 60000, 65536]:
        assert lib.pda_opt_n_threads(n) == oracle.opt_n_threads(n)


def test_wgrad_form_query(lib):
    """Host-side dispatch rule of pda_linear_wgrad (include/pda_train.h): narrow layers stream, wide layers with enough work
    take the split-bf16 kernel, the rest the f32-MFMA split-K kernel; the scratch size follows the same rule."""
    form = lambda t, i, o: lib.pda_linear_wgrad_form(t, i, o)
    assert form(1048576, 32, 64) == 1 and form(8192, 64, 64) == 1
    assert form(131072, 512, 512) == 2 and form(62517, 512, 1536) == 2 and form(32768, 256, 256) == 2
    assert form(12979, 256, 256) == 0          # too little work for 256 KB of partials per workgroup
    assert form(131072, 256, 128) == 0 and form(131072, 260, 256) == 0 and form(4000, 512, 1536) == 0
    for t, i, o in [(131072, 512, 512), (12979, 256, 256), (1048576, 32, 64)]:
        assert lib.pda_linear_wgrad_scratch_bytes(t, i, o) >= (i * o + o) * 4


def test_argument_validation_without_gpu(lib):
    # invalid sizes are rejected before any HIP call, so this runs on a CPU-only box
    st = lib.pda_ball_query(None, None, None, 1, 8, 8, ctypes.c_float(1.0), 0, None)
    assert st == 1 and b"nsample" in lib.pda_last_error()
    st = lib.pda_furthest_point_sampling(None, None, None, -1, 8, 8, None)
    assert st == 1
    st = lib.pda_group_points(None, None, None, 1, 1, 1, -1, 1, None)
    assert st == 1
    # the BatchNorm-folded contraction: sizes, pointer pairing and the statistics mode are checked before anything is launched
    assert lib.pda_gemm_split_bn_tiles(0) == 0 and lib.pda_gemm_split_bn_tiles(1) == 1 and lib.pda_gemm_split_bn_tiles(131072) == 512
    assert lib.pda_gemm_split_bn_tiles(131073) == 513
    assert lib.pda_gemm_split_bn(None, None, None, 0, 256, 256, None, None, None, 0, None, None) == 1 and b"bad size" in lib.pda_last_error()
    assert lib.pda_gemm_split_bn(None, None, None, 4096, 256, 256, None, None, None, 0, None, None) == 1 and b"null" in lib.pda_last_error()
    assert lib.pda_linear_wgrad_bn(None, None, None, None, None, 65536, 256, 256, None, None, None, None) == 1
    assert lib.pda_bn_finalize_fwd(None, 0, 64, 100, ctypes.c_float(1e-5), ctypes.c_float(0.1), None, None, None, None) == 1
    # the entries of ABI 20: sizes are checked before any pointer is used
    i64 = ctypes.c_int64
    assert lib.pda_group_attention_ragged_fwd(None, None, None, None, None, i64(0), i64(0), 32, 4, 64, None) == 0      # no groups
    assert lib.pda_group_attention_ragged_fwd(None, None, None, None, None, i64(-1), i64(8), 32, 4, 64, None) == 1
    assert lib.pda_group_attention_ragged_fwd(None, None, None, None, None, i64(8), i64(8), 33, 4, 64, None) == 1 and b"bad size" in lib.pda_last_error()
    f = ctypes.c_float
    assert lib.pda_densitynet_fwd_unique(None, None, None, None, None, None, None, None, None, None, None, i64(100), None, None, None, 16,
                                         f(1e-5), f(0.1), None) == 1 and b"null" in lib.pda_last_error()
    one = (ctypes.c_int32 * 1)(0)
    onef = (ctypes.c_float * 1)(0)
    assert lib.pda_densitynet_fwd_unique(None, None, None, None, None, None, None, None, None, None, None, i64(100), one, onef, one, 16,
                                         f(1e-5), f(0.1), None) == 1 and b"groups x nsample" in lib.pda_last_error()
    assert lib.pda_densitynet_bwd_unique(None, None, None, None, None, None, i64(96), one, onef, one, 16, f(1e-5), None) == 1 \
        and b"null" in lib.pda_last_error()
    assert lib.pda_assemble_tokens_ragged_grad(None, None, None, None, None, None, None, None, None, None, None, i64(-1), 1, 8, 8, 16, 64, 0,
                                               None) == 1
    assert lib.pda_assemble_tokens_ragged_grad(None, None, None, None, None, None, None, None, None, None, None, i64(0), 1, 8, 0, 16, 64, 0,
                                               None) == 0          # no centres
    # empty problems are PDA_OK and touch nothing
    assert lib.pda_ball_query(None, None, None, 0, 8, 8, ctypes.c_float(1.0), 4, None) == 0
    assert lib.pda_furthest_point_sampling(None, None, None, 2, 8, 0, None) == 0
    assert lib.pda_group_points(None, None, None, 2, 0, 4, 4, 4, None) == 0


def test_mirror_module_matches_reference_names():
    # the 11 reference extension entry points on this path (pointnet2_api.cpp:12-33)
    from pdanet_amd import pointnet2_batch_cuda as ext
    for name in ["ball_query_wrapper", "ball_query_dilated_wrapper", "group_points_wrapper",
                 "group_points_grad_wrapper", "gather_points_wrapper", "gather_points_grad_wrapper",
                 "farthest_point_sampling_wrapper", "furthest_point_sampling_with_dist_wrapper",
                 "three_nn_wrapper", "three_interpolate_wrapper", "three_interpolate_grad_wrapper",
                 "chamfer_forward", "chamfer_backward", "ellipsoid_query"]:
        assert callable(getattr(ext, name))


def test_ellipsoid_query_validates_its_arguments(lib):
    """pointnet2_api.cpp:16: implemented since round 3 (csrc/ellipsoid_query.hip); bad sizes are refused before any HIP call."""
    st = lib.pda_ellipsoid_query(None, None, None, 1, 8, 8, ctypes.c_float(1), ctypes.c_float(2), ctypes.c_float(1), 0, None)
    assert st == 1 and b"pda_ellipsoid_query" in lib.pda_last_error()
    assert lib.pda_ellipsoid_query(None, None, None, 0, 8, 8, ctypes.c_float(1), ctypes.c_float(2), ctypes.c_float(1), 4, None) == 0


def test_ops_refuse_cpu_tensors():
    import torch
    from pdanet_amd import pointnet2_utils as pu
    xyz = torch.zeros(1, 16, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        pu.ball_query(1.0, 4, xyz, xyz)
    with pytest.raises(RuntimeError, match="no CPU path"):
        pu.furthest_point_sample(xyz, 4)
