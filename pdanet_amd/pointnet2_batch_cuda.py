"""Mirror of the reference's extension module ``pointnet2_batch_cuda``.

Same function names and positional signatures as the pybind module defined in
/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:12-33, so the
reference's ``pointnet2_utils.py`` could do ``from pdanet_amd import pointnet2_batch_cuda as
pointnet2`` unchanged.  Each wrapper validates like the reference's CHECK_INPUT
(ball_query.cpp:17-29: CUDA tensor + contiguous), launches the HIP kernel on torch's CURRENT
stream through the C ABI (include/pda_pointnet2.h) and returns what the reference returns
(1, 2 for the with-dist FPS, None for the interpolate trio).  Where the reference prints and
calls exit(-1), this raises.  ``ellipsoid_query`` is not reached by PDA-SSD.yaml and is implemented all the
same (csrc/ellipsoid_query.hip), as are ``chamfer_forward/backward``: the module's name set equals the
reference's and every name has a kernel.
"""
import ctypes

import functools

import torch

from . import _lib


def _chk(t, name, dtype):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA(HIP) tensor -- there is no CPU path" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t.data_ptr()      # a plain int: ctypes converts it for a c_void_p parameter (no wrapper object per argument)


def _numel_ok(t, n, name):
    if t.numel() < n:
        raise RuntimeError("%s has %d elements, the sizes passed need %d" % (name, t.numel(), n))


def _stream(t):
    # raw hipStream_t of torch's current stream on t's device (the C call behind torch.cuda.current_stream(): the
    # Python Stream object costs ~5 us per launch, and a training step makes ~1000 launches through here)
    return torch._C._cuda_getCurrentRawStream(t.device.index)


_FNS = {}


def _call(fn_name, tensor, *args):
    # ~320 calls per training iteration: the bound foreign function is looked up once, the device compared through the
    # C getter, pointers and the stream handle passed as plain ints
    fn = _FNS.get(fn_name)
    if fn is None:
        fn = _FNS[fn_name] = getattr(_lib.load(), fn_name)
    idx = tensor.device.index
    if idx == torch._C._cuda_getDevice():
        st = fn(*args, torch._C._cuda_getCurrentRawStream(idx))
    else:
        with torch.cuda.device(tensor.device):
            st = fn(*args, torch._C._cuda_getCurrentRawStream(idx))
    if st != 0:
        _lib.check(st, fn_name)


F32, I32 = torch.float32, torch.int32


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    _numel_ok(new_xyz, b * m * 3, "new_xyz"); _numel_ok(xyz, b * n * 3, "xyz")
    _numel_ok(idx, b * m * nsample, "idx")
    _call("pda_ball_query", xyz, _chk(new_xyz, "new_xyz", F32), _chk(xyz, "xyz", F32),
          _chk(idx, "idx", I32), b, n, m, float(radius), nsample)
    return 1


def ball_query_dilated_wrapper(b, n, m, max_radius, min_radius, nsample, new_xyz, xyz, idx):
    _numel_ok(new_xyz, b * m * 3, "new_xyz"); _numel_ok(xyz, b * n * 3, "xyz")
    _numel_ok(idx, b * m * nsample, "idx")
    _call("pda_ball_query_dilated", xyz, _chk(new_xyz, "new_xyz", F32), _chk(xyz, "xyz", F32),
          _chk(idx, "idx", I32), b, n, m, float(max_radius), float(min_radius), nsample)
    return 1


def fps_coop_timeouts(reset=False):
    """Exchanges of the multi-workgroup FPS form (n > 24576) that timed out and were recovered on the device since
    the library was loaded / last reset (include/pda_pointnet2.h).  Synchronises the device."""
    total = ctypes.c_ulonglong(0)
    _lib.check(_lib.load().pda_fps_coop_timeouts(ctypes.byref(total), 1 if reset else 0), "pda_fps_coop_timeouts")
    return int(total.value)


def debug_fps_spin_limit(polls):
    _lib.check(_lib.load().pda_debug_fps_spin_limit(int(polls)), "pda_debug_fps_spin_limit")


def ellipsoid_query(new_xyz, xyz, e1, e2, e3, nsample):
    """pointnet2_api.cpp:16 / ellipsoid_query.cpp:13-76: allocates (zero-filled) and returns idx (b, m, nsample): the ball
    query of radius e3 extended by the points inside the ellipsoid (e1, e2, e3) aligned with the hits' principal axes."""
    b, m, n = new_xyz.shape[0], new_xyz.shape[1], xyz.shape[1]
    idx = torch.zeros((b, m, nsample), dtype=I32, device=new_xyz.device)
    _call("pda_ellipsoid_query", xyz, _chk(new_xyz, "new_xyz", F32), _chk(xyz, "xyz", F32), _chk(idx, "idx", I32),
          b, n, m, float(e1), float(e2), float(e3), int(nsample))
    return idx


def ball_query_multi(b, n, m, radii, nsamples, new_xyz, xyz, idxs):
    """MI355X extension: several radii over the same centres/points in one pass."""
    nr = len(radii)
    assert nr == len(nsamples) == len(idxs)
    _numel_ok(new_xyz, b * m * 3, "new_xyz"); _numel_ok(xyz, b * n * 3, "xyz")
    ptrs = (ctypes.c_void_p * nr)()
    for i, (t, ns) in enumerate(zip(idxs, nsamples)):
        _numel_ok(t, b * m * ns, "idx[%d]" % i)
        ptrs[i] = _chk(t, "idx[%d]" % i, I32)
    r = (ctypes.c_float * nr)(*[float(x) for x in radii])
    s = (ctypes.c_int32 * nr)(*[int(x) for x in nsamples])
    _call("pda_ball_query_multi", xyz, _chk(new_xyz, "new_xyz", F32), _chk(xyz, "xyz", F32),
          ptrs, b, n, m, nr, r, s)
    return 1


@functools.lru_cache(maxsize=None)
def ball_query_cells_scratch_bytes(b, n):
    return int(_lib.load().pda_ball_query_cells_scratch_bytes(int(b), int(n)))


def ball_query_cells(b, n, m, radii, nsamples, new_xyz, xyz, idxs, scratch):
    """MI355X extension: ball_query_multi through a uniform cell list (csrc/ball_query_cells.hip); identical rows.
    scratch: uint8 tensor of ball_query_cells_scratch_bytes(b, n) bytes."""
    nr = len(radii)
    assert nr == len(nsamples) == len(idxs)
    _numel_ok(new_xyz, b * m * 3, "new_xyz"); _numel_ok(xyz, b * n * 3, "xyz")
    ptrs = (ctypes.c_void_p * nr)()
    for i, (t, ns) in enumerate(zip(idxs, nsamples)):
        _numel_ok(t, b * m * ns, "idx[%d]" % i)
        ptrs[i] = _chk(t, "idx[%d]" % i, I32)
    r = (ctypes.c_float * nr)(*[float(x) for x in radii])
    s = (ctypes.c_int32 * nr)(*[int(x) for x in nsamples])
    _call("pda_ball_query_cells", xyz, _chk(new_xyz, "new_xyz", F32), _chk(xyz, "xyz", F32), ptrs, b, n, m, nr, r, s,
          _chk(scratch, "scratch", torch.uint8), scratch.numel())
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _numel_ok(points, b * c * n, "points"); _numel_ok(idx, b * npoints * nsample, "idx")
    _numel_ok(out, b * c * npoints * nsample, "out")
    _call("pda_group_points", points, _chk(points, "points", F32), _chk(idx, "idx", I32),
          _chk(out, "out", F32), b, c, n, npoints, nsample)
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _numel_ok(grad_out, b * c * npoints * nsample, "grad_out")
    _numel_ok(idx, b * npoints * nsample, "idx"); _numel_ok(grad_points, b * c * n, "grad_points")
    _call("pda_group_points_grad", grad_out, _chk(grad_out, "grad_out", F32), _chk(idx, "idx", I32),
          _chk(grad_points, "grad_points", F32), b, c, n, npoints, nsample)
    return 1


def group_rows(b, n, c, num_idx, rows, idx, out):
    """MI355X extension: out[b,e,:] = rows[b, idx[b,e], :] (point-major layout)."""
    _numel_ok(rows, b * n * c, "rows"); _numel_ok(idx, b * num_idx, "idx"); _numel_ok(out, b * num_idx * c, "out")
    _call("pda_group_rows", rows, _chk(rows, "rows", F32), _chk(idx, "idx", I32), _chk(out, "out", F32),
          b, n, c, num_idx)
    return 1


def group_rows_grad(b, n, c, num_idx, grad_out, idx, grad_rows):
    _numel_ok(grad_out, b * num_idx * c, "grad_out"); _numel_ok(idx, b * num_idx, "idx")
    _numel_ok(grad_rows, b * n * c, "grad_rows")
    _call("pda_group_rows_grad", grad_out, _chk(grad_out, "grad_out", F32), _chk(idx, "idx", I32),
          _chk(grad_rows, "grad_rows", F32), b, n, c, num_idx)
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _numel_ok(points, b * c * n, "points"); _numel_ok(idx, b * npoints, "idx")
    _numel_ok(out, b * c * npoints, "out")
    _call("pda_gather_points", points, _chk(points, "points", F32), _chk(idx, "idx", I32),
          _chk(out, "out", F32), b, c, n, npoints)
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _numel_ok(grad_out, b * c * npoints, "grad_out"); _numel_ok(idx, b * npoints, "idx")
    _numel_ok(grad_points, b * c * n, "grad_points")
    _call("pda_gather_points_grad", grad_out, _chk(grad_out, "grad_out", F32), _chk(idx, "idx", I32),
          _chk(grad_points, "grad_points", F32), b, c, n, npoints)
    return 1


def farthest_point_sampling_wrapper(b, n, m, xyz, temp, idx):
    _numel_ok(xyz, b * n * 3, "xyz"); _numel_ok(temp, b * n, "temp"); _numel_ok(idx, b * m, "idx")
    _call("pda_furthest_point_sampling", xyz, _chk(xyz, "xyz", F32), _chk(temp, "temp", F32),
          _chk(idx, "idx", I32), b, n, m)
    return 1


def furthest_point_sampling_with_dist_wrapper(b, n, m, dist, temp, idx):
    _numel_ok(dist, b * n * n, "dist"); _numel_ok(temp, b * n, "temp"); _numel_ok(idx, b * m, "idx")
    _call("pda_furthest_point_sampling_with_dist", dist, _chk(dist, "dist", F32),
          _chk(temp, "temp", F32), _chk(idx, "idx", I32), b, n, m)
    return 2  # sampling.cpp:55


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _numel_ok(unknown, b * n * 3, "unknown"); _numel_ok(known, b * m * 3, "known")
    _numel_ok(dist2, b * n * 3, "dist2"); _numel_ok(idx, b * n * 3, "idx")
    _call("pda_three_nn", unknown, _chk(unknown, "unknown", F32), _chk(known, "known", F32),
          _chk(dist2, "dist2", F32), _chk(idx, "idx", I32), b, n, m)


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _numel_ok(points, b * c * m, "points"); _numel_ok(idx, b * n * 3, "idx")
    _numel_ok(weight, b * n * 3, "weight"); _numel_ok(out, b * c * n, "out")
    _call("pda_three_interpolate", points, _chk(points, "points", F32), _chk(idx, "idx", I32),
          _chk(weight, "weight", F32), _chk(out, "out", F32), b, c, m, n)


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _numel_ok(grad_out, b * c * n, "grad_out"); _numel_ok(idx, b * n * 3, "idx")
    _numel_ok(weight, b * n * 3, "weight"); _numel_ok(grad_points, b * c * m, "grad_points")
    _call("pda_three_interpolate_grad", grad_out, _chk(grad_out, "grad_out", F32),
          _chk(idx, "idx", I32), _chk(weight, "weight", F32),
          _chk(grad_points, "grad_points", F32), b, c, n, m)


def chamfer_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """chamfer_cuda.cpp:22-25: tensors only, sizes come from the shapes.  Returns 1."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    _numel_ok(dist1, b * n, "dist1"); _numel_ok(dist2, b * m, "dist2")
    _numel_ok(idx1, b * n, "idx1"); _numel_ok(idx2, b * m, "idx2")
    _call("pda_chamfer_forward", xyz1, _chk(xyz1, "xyz1", F32), _chk(xyz2, "xyz2", F32), _chk(dist1, "dist1", F32),
          _chk(dist2, "dist2", F32), _chk(idx1, "idx1", I32), _chk(idx2, "idx2", I32), b, n, m)
    return 1


def chamfer_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """chamfer_cuda.cpp:27-31.  gradxyz1/2 pre-zeroed by the caller.  Returns 1."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    _call("pda_chamfer_backward", xyz1, _chk(xyz1, "xyz1", F32), _chk(xyz2, "xyz2", F32),
          _chk(gradxyz1, "gradxyz1", F32), _chk(gradxyz2, "gradxyz2", F32), _chk(graddist1, "graddist1", F32),
          _chk(graddist2, "graddist2", F32), _chk(idx1, "idx1", I32), _chk(idx2, "idx2", I32), b, n, m)
    return 1


def group_attention_fwd(qkv, out, lse, num_groups, seq, heads, head_dim):
    """MI355X extension: attention over the tokens of each group (csrc/group_attention.hip).  qkv / out are fp32, or
    both bf16 (dense-bf16 mode: the tensors next to the bf16 GEMMs); lse is always fp32."""
    _numel_ok(qkv, num_groups * seq * 3 * heads * head_dim, "qkv")
    _numel_ok(out, num_groups * seq * heads * head_dim, "out"); _numel_ok(lse, num_groups * heads * seq, "lse")
    io, sfx = (torch.bfloat16, "_bf16") if qkv.dtype == torch.bfloat16 else (F32, "")
    _call("pda_group_attention_fwd" + sfx, qkv, _chk(qkv, "qkv", io), _chk(out, "out", io), _chk(lse, "lse", F32),
          num_groups, seq, heads, head_dim)
    return 1


def group_attention_bwd(qkv, grad_out, lse, grad_qkv, num_groups, seq, heads, head_dim):
    _numel_ok(qkv, num_groups * seq * 3 * heads * head_dim, "qkv"); _numel_ok(grad_qkv, qkv.numel(), "grad_qkv")
    _numel_ok(grad_out, num_groups * seq * heads * head_dim, "grad_out"); _numel_ok(lse, num_groups * heads * seq, "lse")
    io, sfx = (torch.bfloat16, "_bf16") if qkv.dtype == torch.bfloat16 else (F32, "")
    _call("pda_group_attention_bwd" + sfx, qkv, _chk(qkv, "qkv", io), _chk(grad_out, "grad_out", io),
          _chk(lse, "lse", F32), _chk(grad_qkv, "grad_qkv", io), num_groups, seq, heads, head_dim)
    return 1


@functools.lru_cache(maxsize=None)
def bn_relu_scratch_bytes(c):
    return int(_lib.load().pda_bn_relu_scratch_bytes(int(c)))


def _io(t, name):
    """pointer + is-bf16 flag of a tensor that may be fp32 or bf16 (dense-bf16 mode boundary tensors)."""
    if t.dtype == torch.bfloat16:
        return _chk(t, name, torch.bfloat16), 1
    return _chk(t, name, F32), 0


def _buffers_written(running):
    """The kernel about to be launched updates BatchNorm running statistics through raw pointers: tensor version
    counters do not move, so every cache of tensors derived from them (eval-BN folded into convolutions, the packed
    weights + scale/shift of the fused SA kernel) must see a new _lib.PARAM_EPOCH."""
    if running is not None:
        _lib.PARAM_EPOCH[0] += 1


def bn_relu_fwd(x, gamma, beta, running_mean, running_var, y, mean_invstd, scratch, rows, c, eps, momentum):
    """MI355X extension: training-mode BatchNorm + ReLU over the last dim (csrc/bn_relu.hip).  x / y fp32, or bf16 where
    they sit next to a bf16 GEMM (dense-bf16 mode)."""
    _numel_ok(x, rows * c, "x"); _numel_ok(y, rows * c, "y"); _numel_ok(mean_invstd, 2 * c, "mean_invstd")
    rm = None if running_mean is None else _chk(running_mean, "running_mean", F32)
    rv = None if running_var is None else _chk(running_var, "running_var", F32)
    _buffers_written(rm)
    if x.dtype == F32 and y.dtype == F32:
        _call("pda_bn_relu_fwd", x, _chk(x, "x", F32), _chk(gamma, "gamma", F32), _chk(beta, "beta", F32), rm, rv,
              _chk(y, "y", F32), _chk(mean_invstd, "mean_invstd", F32), _chk(scratch, "scratch", torch.uint8), rows, c,
              float(eps), float(momentum))
        return 1
    (xp, xb), (yp, yb) = _io(x, "x"), _io(y, "y")
    _call("pda_bn_relu_fwd_mixed", x, xp, xb, _chk(gamma, "gamma", F32), _chk(beta, "beta", F32), rm, rv, yp, yb,
          _chk(mean_invstd, "mean_invstd", F32), _chk(scratch, "scratch", torch.uint8), rows, c, float(eps), float(momentum))
    return 1


def bn_relu_bwd(x, grad_y, gamma, beta, mean_invstd, grad_x, grad_gamma, grad_beta, scratch, rows, c):
    """grad_x has the dtype of x; grad_y may be fp32 or bf16."""
    _numel_ok(x, rows * c, "x"); _numel_ok(grad_y, rows * c, "grad_y"); _numel_ok(grad_x, rows * c, "grad_x")
    if x.dtype == F32 and grad_y.dtype == F32:
        _call("pda_bn_relu_bwd", x, _chk(x, "x", F32), _chk(grad_y, "grad_y", F32), _chk(gamma, "gamma", F32),
              _chk(beta, "beta", F32), _chk(mean_invstd, "mean_invstd", F32), _chk(grad_x, "grad_x", F32),
              _chk(grad_gamma, "grad_gamma", F32), _chk(grad_beta, "grad_beta", F32), _chk(scratch, "scratch", torch.uint8), rows, c)
        return 1
    (xp, xb), (gp, gb) = _io(x, "x"), _io(grad_y, "grad_y")
    _call("pda_bn_relu_bwd_mixed", x, xp, xb, gp, gb, _chk(gamma, "gamma", F32), _chk(beta, "beta", F32),
          _chk(mean_invstd, "mean_invstd", F32), _chk(grad_x, "grad_x", x.dtype), _chk(grad_gamma, "grad_gamma", F32),
          _chk(grad_beta, "grad_beta", F32), _chk(scratch, "scratch", torch.uint8), rows, c)
    return 1


def bn_relu_max_pool_fwd(x, gamma, beta, running_mean, running_var, out, arg, mean_invstd, scratch, groups, ns, c, eps, momentum):
    """MI355X extension: relu(bn(x)) (training mode) + max over the ns rows of each group, y never written (csrc/bn_relu.hip)."""
    _numel_ok(x, groups * ns * c, "x"); _numel_ok(out, groups * c, "out"); _numel_ok(arg, groups * c, "arg"); _numel_ok(mean_invstd, 2 * c, "mean_invstd")
    rm = None if running_mean is None else _chk(running_mean, "running_mean", F32)
    rv = None if running_var is None else _chk(running_var, "running_var", F32)
    _buffers_written(rm)
    xp, xb = _io(x, "x")
    _call("pda_bn_relu_max_pool_fwd", x, xp, xb, _chk(gamma, "gamma", F32), _chk(beta, "beta", F32), rm, rv, _chk(out, "out", F32),
          _chk(arg, "arg", torch.uint8), _chk(mean_invstd, "mean_invstd", F32), _chk(scratch, "scratch", torch.uint8), groups, ns, c,
          float(eps), float(momentum))
    return 1


def bn_relu_max_pool_bwd(x, grad_out, arg, gamma, beta, mean_invstd, grad_x, grad_gamma, grad_beta, scratch, groups, ns, c):
    _numel_ok(x, groups * ns * c, "x"); _numel_ok(grad_out, groups * c, "grad_out"); _numel_ok(arg, groups * c, "arg")
    _numel_ok(grad_x, groups * ns * c, "grad_x")
    xp, xb = _io(x, "x")
    _call("pda_bn_relu_max_pool_bwd", x, xp, xb, _chk(grad_out, "grad_out", F32), _chk(arg, "arg", torch.uint8), _chk(gamma, "gamma", F32),
          _chk(beta, "beta", F32), _chk(mean_invstd, "mean_invstd", F32), _chk(grad_x, "grad_x", x.dtype), _chk(grad_gamma, "grad_gamma", F32),
          _chk(grad_beta, "grad_beta", F32), _chk(scratch, "scratch", torch.uint8), groups, ns, c)
    return 1


@functools.lru_cache(maxsize=None)
def layer_norm_scratch_bytes(d):
    return int(_lib.load().pda_layer_norm_scratch_bytes(int(d)))


BF16 = torch.bfloat16


def layer_norm_fwd(x, residual, gamma, beta, sum_out, y, mean_rstd, rows, d, eps, y_bf16=None):
    """MI355X extension: LayerNorm over the last dim with optional fused residual add (csrc/layer_norm.hip).
    Dense-bf16 mode: x may be bf16 (a GEMM output), y may be None when only the bf16 copy y_bf16 is wanted."""
    _numel_ok(x, rows * d, "x"); _numel_ok(mean_rstd, rows * 2, "mean_rstd")
    res = None if residual is None else _chk(residual, "residual", F32)
    so = None if sum_out is None else _chk(sum_out, "sum_out", F32)
    if x.dtype == F32 and y_bf16 is None:
        _numel_ok(y, rows * d, "y")
        _call("pda_layer_norm_fwd", x, _chk(x, "x", F32), res, _chk(gamma, "gamma", F32), _chk(beta, "beta", F32), so,
              _chk(y, "y", F32), _chk(mean_rstd, "mean_rstd", F32), rows, d, float(eps))
        return 1
    yp = yb = None
    if y is not None:
        _numel_ok(y, rows * d, "y"); yp = _chk(y, "y", F32)
    if y_bf16 is not None:
        _numel_ok(y_bf16, rows * d, "y_bf16"); yb = _chk(y_bf16, "y_bf16", BF16)
    _call("pda_layer_norm_fwd_mixed", x, _chk(x, "x", x.dtype if x.dtype == BF16 else F32), int(x.dtype == BF16), res,
          _chk(gamma, "gamma", F32), _chk(beta, "beta", F32), so, yp, yb, _chk(mean_rstd, "mean_rstd", F32), rows, d, float(eps))
    return 1


def layer_norm_bwd(x, grad_y, gamma, mean_rstd, grad_x, grad_gamma, grad_beta, scratch, rows, d, grad_y2=None, grad_x_bf16=None):
    """grad_y2: optional second incoming gradient (fp32, or bf16 in dense-bf16 mode); grad_x_bf16: optional bf16 copy of grad_x."""
    _numel_ok(x, rows * d, "x"); _numel_ok(grad_y, rows * d, "grad_y"); _numel_ok(grad_x, rows * d, "grad_x")
    g2 = None
    if grad_y2 is not None:
        _numel_ok(grad_y2, rows * d, "grad_y2")
        g2 = _chk(grad_y2, "grad_y2", BF16 if grad_y2.dtype == BF16 else F32)
    if grad_x_bf16 is not None or (grad_y2 is not None and grad_y2.dtype == BF16):
        gxb = None
        if grad_x_bf16 is not None:
            _numel_ok(grad_x_bf16, rows * d, "grad_x_bf16"); gxb = _chk(grad_x_bf16, "grad_x_bf16", BF16)
        _call("pda_layer_norm_bwd_mixed", x, _chk(x, "x", F32), _chk(grad_y, "grad_y", F32), g2,
              int(grad_y2 is not None and grad_y2.dtype == BF16), _chk(gamma, "gamma", F32), _chk(mean_rstd, "mean_rstd", F32),
              _chk(grad_x, "grad_x", F32), gxb, _chk(grad_gamma, "grad_gamma", F32), _chk(grad_beta, "grad_beta", F32),
              _chk(scratch, "scratch", torch.uint8), rows, d)
        return 1
    _call("pda_layer_norm_bwd", x, _chk(x, "x", F32), _chk(grad_y, "grad_y", F32), g2, _chk(gamma, "gamma", F32),
          _chk(mean_rstd, "mean_rstd", F32), _chk(grad_x, "grad_x", F32), _chk(grad_gamma, "grad_gamma", F32),
          _chk(grad_beta, "grad_beta", F32), _chk(scratch, "scratch", torch.uint8), rows, d)
    return 1


_WGRAD_SCRATCH_BYTES = {}


def _wgrad_scratch_bytes(tokens, in_features, out_features):
    # ~60 weight gradients per training step: the size query is a foreign call of its own, so it is cached -- boundedly: the
    # token counts of the unique-token encoder change every step, and a table keyed on them would grow for as long as a run lasts
    key = (tokens, in_features, out_features)
    nbytes = _WGRAD_SCRATCH_BYTES.get(key)
    if nbytes is None:
        if len(_WGRAD_SCRATCH_BYTES) >= 4096:
            _WGRAD_SCRATCH_BYTES.clear()
        nbytes = _WGRAD_SCRATCH_BYTES[key] = int(_lib.load().pda_linear_wgrad_scratch_bytes(tokens, in_features, out_features))
    return nbytes


@functools.lru_cache(maxsize=None)
def _split_packed_bytes(n_out, k):
    """Size queries are pure functions of the shape: cached, a training step asks ~300 times (each a foreign call)."""
    return int(_lib.load().pda_linear_split_packed_bytes(n_out, k))


@functools.lru_cache(maxsize=None)
def _cols_packed_size(n_out, k):
    return int(_lib.load().pda_linear_cols_packed_size(n_out, k))


def linear_wgrad(x, grad_out, grad_weight, grad_bias, tokens, in_features, out_features):
    """MI355X extension: grad_weight (out, in) = grad_out^T x and grad_bias = column sums (csrc/wgrad.hip)."""
    _numel_ok(x, tokens * in_features, "x"); _numel_ok(grad_out, tokens * out_features, "grad_out")
    _numel_ok(grad_weight, in_features * out_features, "grad_weight")
    nbytes = _wgrad_scratch_bytes(tokens, in_features, out_features)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    gb = None if grad_bias is None else _chk(grad_bias, "grad_bias", F32)
    _call("pda_linear_wgrad", x, _chk(x, "x", F32), _chk(grad_out, "grad_out", F32), _chk(grad_weight, "grad_weight", F32),
          gb, _chk(scratch, "scratch", torch.uint8), tokens, in_features, out_features)
    return 1


def linear_wgrad_bn(x, grad_out, grad_weight, tokens, in_features, out_features, x_mean_invstd, x_gamma, x_beta):
    """grad_weight (out, in) = grad_out^T relu(bn(x)) with the BatchNorm + ReLU of the PRE-BatchNorm tensor x formed in the
    operand load (csrc/wgrad.hip, split form only: PdaError otherwise)."""
    _numel_ok(x, tokens * in_features, "x"); _numel_ok(grad_out, tokens * out_features, "grad_out")
    _numel_ok(grad_weight, in_features * out_features, "grad_weight"); _numel_ok(x_mean_invstd, 2 * in_features, "x_mean_invstd")
    _numel_ok(x_gamma, in_features, "x_gamma"); _numel_ok(x_beta, in_features, "x_beta")
    nbytes = _wgrad_scratch_bytes(tokens, in_features, out_features)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    _call("pda_linear_wgrad_bn", x, _chk(x, "x", F32), _chk(grad_out, "grad_out", F32), _chk(grad_weight, "grad_weight", F32), None,
          _chk(scratch, "scratch", torch.uint8), tokens, in_features, out_features, _chk(x_mean_invstd, "x_mean_invstd", F32),
          _chk(x_gamma, "x_gamma", F32), _chk(x_beta, "x_beta", F32))
    return 1


def gemm_split_bn_tiles(tokens):
    return (int(tokens) + 255) // 256          # == pda_gemm_split_bn_tiles (tests/test_capi_symbols.py)


def gemm_split_bn(x, wf, y, tokens, k, n_out, in_bn=None, stats_mode=0, partial=None):
    """y = X' W^T on the 256 x 256 tile split-bf16 kernel; X' = relu(bn(x)) when in_bn = (mean_invstd, gamma, beta) of the input
    channels; stats_mode 1: `partial` [tiles][2][n_out] (float64) receives the per-column sums of y (include/pda_train.h)."""
    _numel_ok(x, tokens * k, "x"); _numel_ok(y, tokens * n_out, "y")
    _numel_ok(wf, _split_packed_bytes(int(n_out), int(k)), "wf")
    ib = (None, None, None)
    if in_bn is not None:
        _numel_ok(in_bn[0], 2 * k, "in_mean_invstd"); _numel_ok(in_bn[1], k, "in_gamma"); _numel_ok(in_bn[2], k, "in_beta")
        ib = tuple(_chk(t, "in_bn", F32) for t in in_bn)
    if stats_mode:
        _numel_ok(partial, gemm_split_bn_tiles(tokens) * 2 * n_out, "partial")
    _call("pda_gemm_split_bn", x, _chk(x, "x", F32), _chk(wf, "wf", torch.uint8), _chk(y, "y", F32), tokens, k, n_out, ib[0], ib[1], ib[2],
          int(stats_mode), None if partial is None else _chk(partial, "partial", torch.float64))
    return 1


def gemm_split_maxpool(x, wf, bias, out, tokens, k, n_out, ns, relu=True):
    """out (tokens / ns, n_out) = max over each group of ns consecutive rows of relu?(x W^T + bias) (csrc/gemm_split.hip)."""
    _numel_ok(x, tokens * k, "x"); _numel_ok(out, (tokens // ns) * n_out, "out")
    _numel_ok(wf, _split_packed_bytes(int(n_out), int(k)), "wf")
    if bias is not None:
        _numel_ok(bias, n_out, "bias")
    _call("pda_gemm_split_maxpool", x, _chk(x, "x", F32), _chk(wf, "wf", torch.uint8), None if bias is None else _chk(bias, "bias", F32),
          _chk(out, "out", F32), tokens, k, n_out, ns, 1 if relu else 0)
    return 1


def gemm_split_gather(point_rows, xyz, new_xyz, idx, w1, bias1, wf, bias2, y, b, n, m, ns, k, n_out, relu=True):
    """y (b*m*ns, n_out) = relu?(A W2^T + bias2), A = the first SA layer's output formed in the operand load from the per-point
    projection `point_rows` (b*n, k), the neighbour lists and the coordinate columns of w1 (k, 3 + c) (include/pda_train.h)."""
    t = b * m * ns
    _numel_ok(point_rows, b * n * k, "point_rows"); _numel_ok(xyz, b * n * 3, "xyz"); _numel_ok(new_xyz, b * m * 3, "new_xyz")
    _numel_ok(idx, t, "idx"); _numel_ok(y, t * n_out, "y"); _numel_ok(wf, _split_packed_bytes(int(n_out), int(k)), "wf")
    if w1.dim() != 2 or w1.shape[0] != k or w1.shape[1] < 3:
        raise RuntimeError("w1 must be (k, 3 + c)")
    _call("pda_gemm_split_gather", y, _chk(point_rows, "point_rows", F32), _chk(xyz, "xyz", F32), _chk(new_xyz, "new_xyz", F32),
          _chk(idx, "idx", I32), _chk(w1, "w1", F32), int(w1.shape[1]), None if bias1 is None else _chk(bias1, "bias1", F32),
          _chk(wf, "wf", torch.uint8), None if bias2 is None else _chk(bias2, "bias2", F32), _chk(y, "y", F32), b, n, m, ns, k, n_out,
          1 if relu else 0)
    return 1


def bn_stats_fwd(x, running_mean, running_var, mean_invstd, scratch, rows, c, eps, momentum):
    """The statistics pass of bn_relu_fwd alone (mean_invstd (2, C); running statistics updated)."""
    _numel_ok(x, rows * c, "x"); _numel_ok(mean_invstd, 2 * c, "mean_invstd")
    rm = None if running_mean is None else _chk(running_mean, "running_mean", F32)
    rv = None if running_var is None else _chk(running_var, "running_var", F32)
    _buffers_written(rm)
    _call("pda_bn_stats_fwd", x, _chk(x, "x", F32), rm, rv, _chk(mean_invstd, "mean_invstd", F32), _chk(scratch, "scratch", torch.uint8),
          rows, c, float(eps), float(momentum))
    return 1


def bn_finalize_fwd(partial, nblocks, c, count, eps, momentum, mean_invstd, running_mean, running_var):
    _numel_ok(partial, nblocks * 2 * c, "partial"); _numel_ok(mean_invstd, 2 * c, "mean_invstd")
    rm = None if running_mean is None else _chk(running_mean, "running_mean", F32)
    rv = None if running_var is None else _chk(running_var, "running_var", F32)
    _buffers_written(rm)
    _call("pda_bn_finalize_fwd", partial, _chk(partial, "partial", torch.float64), nblocks, c, count, float(eps), float(momentum),
          _chk(mean_invstd, "mean_invstd", F32), rm, rv)
    return 1


def bn_relu_max_pool_apply(x, gamma, beta, mean_invstd, out, arg, groups, ns, c):
    _numel_ok(x, groups * ns * c, "x"); _numel_ok(out, groups * c, "out"); _numel_ok(arg, groups * c, "arg"); _numel_ok(mean_invstd, 2 * c, "mean_invstd")
    _call("pda_bn_relu_max_pool_apply", x, _chk(x, "x", F32), _chk(gamma, "gamma", F32), _chk(beta, "beta", F32),
          _chk(mean_invstd, "mean_invstd", F32), _chk(out, "out", F32), _chk(arg, "arg", torch.uint8), groups, ns, c)
    return 1


def colsum_bf16(g, out, rows, cols):
    """MI355X extension: out (cols) fp32 = column sums of the bf16 matrix g (rows, cols): the bias gradient in dense-bf16
    mode (csrc/wgrad.hip; fixed summation order)."""
    _numel_ok(g, rows * cols, "g"); _numel_ok(out, cols, "out")
    scratch = torch.empty((int(_lib.load().pda_colsum_scratch_bytes(cols)),), dtype=torch.uint8, device=g.device)
    _call("pda_colsum_bf16", g, _chk(g, "g", torch.bfloat16), _chk(out, "out", F32), _chk(scratch, "scratch", torch.uint8), rows, cols)
    return 1


@functools.lru_cache(maxsize=None)
def densitynet_sizes():
    lib = _lib.load()
    return int(lib.pda_densitynet_param_count()), int(lib.pda_densitynet_scratch_bytes())


def densitynet_eval(x, folded, y, n):
    """MI355X extension: inference DensityNet, BatchNorm folded into the three layers, one launch (csrc/densitynet.hip).
    folded: 177 floats w1[16] b1[16] W2[8][16] b2[8] w3[8] b3."""
    _numel_ok(x, n, "x"); _numel_ok(y, n, "y"); _numel_ok(folded, int(_lib.load().pda_densitynet_eval_param_count()), "folded")
    _call("pda_densitynet_eval", x, _chk(x, "x", F32), _chk(folded, "folded", F32), _chk(y, "y", F32), n)
    return 1


def densitynet_fwd(x, params, y, stats, scratch, running, n, eps, momentum):
    """MI355X extension: training-mode DensityNet on a scalar input per token (csrc/densitynet.hip).
    running: [rm1, rv1, rm2, rv2, rm3, rv3] or None."""
    _numel_ok(x, n, "x"); _numel_ok(y, n, "y")
    r = [None] * 6 if running is None else [_chk(t, "running", F32) for t in running]
    _buffers_written(running)
    _call("pda_densitynet_fwd", x, _chk(x, "x", F32), _chk(params, "params", F32), _chk(y, "y", F32), _chk(stats, "stats", F32),
          _chk(scratch, "scratch", torch.uint8), *r, n, float(eps), float(momentum))
    return 1


def densitynet_bwd(x, grad_y, params, stats, grad_params, scratch, n, eps):
    _numel_ok(x, n, "x"); _numel_ok(grad_y, n, "grad_y")
    _call("pda_densitynet_bwd", x, _chk(x, "x", F32), _chk(grad_y, "grad_y", F32), _chk(params, "params", F32),
          _chk(stats, "stats", F32), _chk(grad_params, "grad_params", F32), _chk(scratch, "scratch", torch.uint8), n, float(eps))
    return 1


def _unique_rows(rowmap, roww, off, groups, n, nsample):
    """(rowmap, row_weight, n_unique) pointers of a plan (pda_ragged_plan): n_unique = &off[groups], on the device."""
    assert groups * nsample == n and off.numel() >= groups + 1 and off.dtype == I32 and off.is_contiguous()
    _numel_ok(rowmap, 0, "rowmap"); _numel_ok(roww, 0, "row_weight")
    return _chk(rowmap, "rowmap", I32), _chk(roww, "row_weight", F32), off.data_ptr() + 4 * groups


def densitynet_fwd_unique(x, params, y, stats, scratch, running, n, rowmap, roww, off, groups, nsample, eps, momentum):
    """densitynet_fwd on the distinct slots of padded neighbour lists (include/pda_train.h): the same y in every slot."""
    _numel_ok(x, n, "x"); _numel_ok(y, n, "y")
    r = [None] * 6 if running is None else [_chk(t, "running", F32) for t in running]
    _buffers_written(running)
    rm, rw, nu = _unique_rows(rowmap, roww, off, groups, n, nsample)
    _call("pda_densitynet_fwd_unique", x, _chk(x, "x", F32), _chk(params, "params", F32), _chk(y, "y", F32), _chk(stats, "stats", F32),
          _chk(scratch, "scratch", torch.uint8), *r, n, rm, rw, nu, nsample, float(eps), float(momentum))
    return 1


def densitynet_bwd_unique(x, grad_y, params, stats, grad_params, scratch, n, rowmap, roww, off, groups, nsample, eps):
    _numel_ok(x, n, "x"); _numel_ok(grad_y, n, "grad_y")
    rm, rw, nu = _unique_rows(rowmap, roww, off, groups, n, nsample)
    _call("pda_densitynet_bwd_unique", x, _chk(x, "x", F32), _chk(grad_y, "grad_y", F32), _chk(params, "params", F32),
          _chk(stats, "stats", F32), _chk(grad_params, "grad_params", F32), _chk(scratch, "scratch", torch.uint8), n, rm, rw, nu,
          nsample, float(eps))
    return 1


def densitynet_multi(problems, backward=False):
    """Several DensityNet problems in one set of launches (pda_densitynet_{fwd,bwd}_multi).  problems: dicts with x, params,
    stats, scratch, n, eps, momentum and y + running (forward) or grad_y + grad_params (backward); optional unique rows:
    rowmap, roww, off, groups, nsample (see densitynet_fwd_unique)."""
    arr = (_lib.DensityNetScale * len(problems))()
    for S, p in zip(arr, problems):
        n = int(p["n"])
        _numel_ok(p["x"], n, "x")
        S.x = _chk(p["x"], "x", F32); S.params = _chk(p["params"], "params", F32); S.stats = _chk(p["stats"], "stats", F32)
        S.scratch = _chk(p["scratch"], "scratch", torch.uint8)
        S.n, S.nsample, S.eps, S.momentum = n, 1, float(p["eps"]), float(p.get("momentum") or 0.0)
        if backward:
            _numel_ok(p["grad_y"], n, "grad_y")
            S.grad_y = _chk(p["grad_y"], "grad_y", F32); S.grad_params = _chk(p["grad_params"], "grad_params", F32)
        else:
            _numel_ok(p["y"], n, "y")
            S.y = _chk(p["y"], "y", F32)
            if p.get("running") is not None:
                _buffers_written(p["running"])
                for k, t in enumerate(p["running"]):
                    S.running[k] = _chk(t, "running", F32)
        if p.get("rowmap") is not None:
            S.rowmap, S.row_weight, S.n_unique = _unique_rows(p["rowmap"], p["roww"], p["off"], p["groups"], n, p["nsample"])
            S.nsample = int(p["nsample"])
    _call("pda_densitynet_bwd_multi" if backward else "pda_densitynet_fwd_multi", problems[0]["x"], ctypes.addressof(arr), len(problems))
    return 1


def pda_geometry(xyz, new_xyz, idx, rppe, dscale, b, n, m, nsample, radius):
    """MI355X extension: relative-position input and normalised gaussian density of a PDA scale (csrc/densitynet.hip)."""
    _numel_ok(xyz, b * n * 3, "xyz"); _numel_ok(new_xyz, b * m * 3, "new_xyz"); _numel_ok(idx, b * m * nsample, "idx")
    _numel_ok(rppe, b * m * nsample * 12, "rppe"); _numel_ok(dscale, b * m * nsample, "dscale")
    _call("pda_pda_geometry", xyz, _chk(xyz, "xyz", F32), _chk(new_xyz, "new_xyz", F32), _chk(idx, "idx", I32),
          _chk(rppe, "rppe", F32), _chk(dscale, "dscale", F32), b, n, m, nsample, float(radius))
    return 1


def add_max_pool(a, b, out, arg, groups, seq, d):
    """MI355X extension: out = max over the seq tokens of a + b, arg = first arg-max token (csrc/layer_norm.hip)."""
    _numel_ok(a, groups * seq * d, "a"); _numel_ok(b, groups * seq * d, "b"); _numel_ok(out, groups * d, "out"); _numel_ok(arg, groups * d, "arg")
    io, sfx = (BF16, "_bf16") if b.dtype == BF16 else (F32, "")      # b: a GEMM output, bf16 in dense-bf16 mode
    _call("pda_add_max_pool" + sfx, a, _chk(a, "a", F32), _chk(b, "b", io), _chk(out, "out", F32), _chk(arg, "arg", torch.uint8), groups, seq, d)
    return 1


def max_pool_scatter(grad_out, arg, grad_x, groups, seq, d, grad_x_bf16=None):
    _numel_ok(grad_out, groups * d, "grad_out"); _numel_ok(arg, groups * d, "arg"); _numel_ok(grad_x, groups * seq * d, "grad_x")
    if grad_x_bf16 is not None:
        _numel_ok(grad_x_bf16, groups * seq * d, "grad_x_bf16")
        _call("pda_max_pool_scatter_bf16", grad_out, _chk(grad_out, "grad_out", F32), _chk(arg, "arg", torch.uint8), _chk(grad_x, "grad_x", F32),
              _chk(grad_x_bf16, "grad_x_bf16", BF16), groups, seq, d)
        return 1
    _call("pda_max_pool_scatter", grad_out, _chk(grad_out, "grad_out", F32), _chk(arg, "arg", torch.uint8), _chk(grad_x, "grad_x", F32),
          groups, seq, d)
    return 1


def assemble_tokens(rppe, dscale, feats, idx, glob, out, b, n, m, nsample, c):
    """MI355X extension: [rppe | f*dscale | f | glob] per (centre, neighbour) token (csrc/assemble.hip)."""
    t = b * m * nsample
    _numel_ok(rppe, t * c, "rppe"); _numel_ok(dscale, t, "dscale"); _numel_ok(feats, b * n * c, "feats"); _numel_ok(idx, t, "idx")
    _numel_ok(glob, b * m * c, "glob"); _numel_ok(out, t * 4 * c, "out")
    _call("pda_assemble_tokens", rppe, _chk(rppe, "rppe", F32), _chk(dscale, "dscale", F32), _chk(feats, "feats", F32),
          _chk(idx, "idx", I32), _chk(glob, "glob", F32), _chk(out, "out", F32), b, n, m, nsample, c)
    return 1


def assemble_tokens_grad(grad_out, dscale, feats, idx, grad_rppe, grad_dscale, grad_feats, grad_glob, b, n, m, nsample, c):
    t = b * m * nsample
    _numel_ok(grad_out, t * 4 * c, "grad_out"); _numel_ok(grad_rppe, t * c, "grad_rppe"); _numel_ok(grad_dscale, t, "grad_dscale")
    _numel_ok(grad_feats, b * n * c, "grad_feats"); _numel_ok(grad_glob, b * m * c, "grad_glob")
    _call("pda_assemble_tokens_grad", grad_out, _chk(grad_out, "grad_out", F32), _chk(dscale, "dscale", F32), _chk(feats, "feats", F32),
          _chk(idx, "idx", I32), _chk(grad_rppe, "grad_rppe", F32), _chk(grad_dscale, "grad_dscale", F32),
          _chk(grad_feats, "grad_feats", F32), _chk(grad_glob, "grad_glob", F32), b, n, m, nsample, c)
    return 1


# ---- unique-token ("ragged") execution of a PDA scale (include/pda_train.h, csrc/ragged.hip) --------------------
def ragged_plan(idx, cnt, off, rowmap, groups, nsample, roww=None):
    """idx (groups, nsample) -> cnt (groups), off (groups + 1), rowmap and (optional) roww, the multiplicity of each compact
    token (groups * nsample capacity).  No synchronisation."""
    _numel_ok(idx, groups * nsample, "idx"); _numel_ok(cnt, groups, "cnt"); _numel_ok(off, groups + 1, "off")
    _numel_ok(rowmap, groups * nsample, "rowmap")
    if roww is not None:
        _numel_ok(roww, groups * nsample, "roww")
    _call("pda_ragged_plan", idx, _chk(idx, "idx", I32), _chk(cnt, "cnt", I32), _chk(off, "off", I32), _chk(rowmap, "rowmap", I32),
          None if roww is None else _chk(roww, "roww", F32), groups, nsample)
    return 1


def bn_relu_fwd_weighted(x, gamma, beta, running_mean, running_var, y, mean_invstd, scratch, rows, c, eps, momentum, roww, count):
    """MI355X extension: bn_relu_fwd on rows that stand for roww[r] identical rows of a dense tensor of `count` rows."""
    _numel_ok(x, rows * c, "x"); _numel_ok(y, rows * c, "y"); _numel_ok(mean_invstd, 2 * c, "mean_invstd"); _numel_ok(roww, rows, "roww")
    rm = None if running_mean is None else _chk(running_mean, "running_mean", F32)
    rv = None if running_var is None else _chk(running_var, "running_var", F32)
    _buffers_written(rm)
    _call("pda_bn_relu_fwd_weighted", x, _chk(x, "x", F32), _chk(gamma, "gamma", F32), _chk(beta, "beta", F32), rm, rv, _chk(y, "y", F32),
          _chk(mean_invstd, "mean_invstd", F32), _chk(scratch, "scratch", torch.uint8), rows, c, float(eps), float(momentum),
          _chk(roww, "roww", F32), int(count))
    return 1


def bn_relu_bwd_weighted(x, grad_y, gamma, beta, mean_invstd, grad_x, grad_gamma, grad_beta, scratch, rows, c, roww, count):
    _numel_ok(x, rows * c, "x"); _numel_ok(grad_y, rows * c, "grad_y"); _numel_ok(grad_x, rows * c, "grad_x"); _numel_ok(roww, rows, "roww")
    _call("pda_bn_relu_bwd_weighted", x, _chk(x, "x", F32), _chk(grad_y, "grad_y", F32), _chk(gamma, "gamma", F32), _chk(beta, "beta", F32),
          _chk(mean_invstd, "mean_invstd", F32), _chk(grad_x, "grad_x", F32), _chk(grad_gamma, "grad_gamma", F32),
          _chk(grad_beta, "grad_beta", F32), _chk(scratch, "scratch", torch.uint8), rows, c, _chk(roww, "roww", F32), int(count))
    return 1


def assemble_tokens_ragged(rppe, dscale, feats, idx, glob, rowmap, off, out, tokens, b, n, m, nsample, c, rppe_compact=False):
    t = b * m * nsample
    _numel_ok(rppe, (tokens if rppe_compact else t) * c, "rppe"); _numel_ok(dscale, t, "dscale"); _numel_ok(feats, b * n * c, "feats"); _numel_ok(idx, t, "idx")
    _numel_ok(glob, b * m * c, "glob"); _numel_ok(out, tokens * 4 * c, "out"); _numel_ok(rowmap, tokens, "rowmap"); _numel_ok(off, b * m + 1, "off")
    _call("pda_assemble_tokens_ragged", rppe, _chk(rppe, "rppe", F32), _chk(dscale, "dscale", F32), _chk(feats, "feats", F32),
          _chk(idx, "idx", I32), _chk(glob, "glob", F32), _chk(rowmap, "rowmap", I32), _chk(off, "off", I32), _chk(out, "out", F32),
          tokens, b, n, m, nsample, c, 1 if rppe_compact else 0)
    return 1


def assemble_tokens_ragged_grad(grad_out, dscale, feats, idx, cnt, off, grad_rppe, grad_dscale, grad_feats, grad_glob, tokens, b, n, m, nsample, c,
                                rppe_compact=False, rowmap=None):
    """rowmap (the plan's compact-row -> dense-slot map, >= tokens entries): the per-token part runs token-parallel."""
    t = b * m * nsample
    if rowmap is not None:
        _numel_ok(rowmap, tokens, "rowmap")
    _numel_ok(grad_out, tokens * 4 * c, "grad_out"); _numel_ok(grad_rppe, (tokens if rppe_compact else t) * c, "grad_rppe"); _numel_ok(grad_dscale, t, "grad_dscale")
    _numel_ok(grad_feats, b * n * c, "grad_feats"); _numel_ok(grad_glob, b * m * c, "grad_glob"); _numel_ok(cnt, b * m, "cnt"); _numel_ok(off, b * m + 1, "off")
    _call("pda_assemble_tokens_ragged_grad", grad_out, _chk(grad_out, "grad_out", F32), _chk(dscale, "dscale", F32), _chk(feats, "feats", F32),
          _chk(idx, "idx", I32), _chk(cnt, "cnt", I32), _chk(off, "off", I32), None if rowmap is None else _chk(rowmap, "rowmap", I32),
          _chk(grad_rppe, "grad_rppe", F32), _chk(grad_dscale, "grad_dscale", F32), _chk(grad_feats, "grad_feats", F32),
          _chk(grad_glob, "grad_glob", F32), tokens, b, n, m, nsample, c, 1 if rppe_compact else 0)
    return 1


def add_max_pool_ragged(a, b, cnt, off, out, arg, tokens, groups, d):
    _numel_ok(a, tokens * d, "a"); _numel_ok(b, tokens * d, "b"); _numel_ok(out, groups * d, "out"); _numel_ok(arg, groups * d, "arg")
    _numel_ok(cnt, groups, "cnt"); _numel_ok(off, groups + 1, "off")
    _call("pda_add_max_pool_ragged", a, _chk(a, "a", F32), _chk(b, "b", F32), _chk(cnt, "cnt", I32), _chk(off, "off", I32),
          _chk(out, "out", F32), _chk(arg, "arg", torch.uint8), groups, d)
    return 1


def max_pool_scatter_ragged(grad_out, arg, rowmap, off, grad_x, tokens, groups, nsample, d):
    _numel_ok(grad_out, groups * d, "grad_out"); _numel_ok(arg, groups * d, "arg"); _numel_ok(grad_x, tokens * d, "grad_x")
    _numel_ok(rowmap, tokens, "rowmap"); _numel_ok(off, groups + 1, "off")
    _call("pda_max_pool_scatter_ragged", grad_out, _chk(grad_out, "grad_out", F32), _chk(arg, "arg", torch.uint8),
          _chk(rowmap, "rowmap", I32), _chk(off, "off", I32), _chk(grad_x, "grad_x", F32), tokens, groups, nsample, d)
    return 1


def group_attention_ragged_fwd(qkv, cnt, off, out, lse, tokens, num_groups, seq, heads, head_dim):
    _numel_ok(qkv, tokens * 3 * heads * head_dim, "qkv"); _numel_ok(out, tokens * heads * head_dim, "out")
    _numel_ok(lse, num_groups * heads * seq, "lse"); _numel_ok(cnt, num_groups, "cnt"); _numel_ok(off, num_groups + 1, "off")
    _call("pda_group_attention_ragged_fwd", qkv, _chk(qkv, "qkv", F32), _chk(cnt, "cnt", I32), _chk(off, "off", I32),
          _chk(out, "out", F32), _chk(lse, "lse", F32), tokens, num_groups, seq, heads, head_dim)
    return 1


def group_attention_ragged_bwd(qkv, grad_out, lse, cnt, off, grad_qkv, tokens, num_groups, seq, heads, head_dim):
    _numel_ok(qkv, tokens * 3 * heads * head_dim, "qkv"); _numel_ok(grad_qkv, tokens * 3 * heads * head_dim, "grad_qkv")
    _numel_ok(grad_out, tokens * heads * head_dim, "grad_out"); _numel_ok(lse, num_groups * heads * seq, "lse")
    _call("pda_group_attention_ragged_bwd", qkv, _chk(qkv, "qkv", F32), _chk(grad_out, "grad_out", F32), _chk(lse, "lse", F32),
          _chk(cnt, "cnt", I32), _chk(off, "off", I32), _chk(grad_qkv, "grad_qkv", F32), tokens, num_groups, seq, heads, head_dim)
    return 1


# ---- training form of the vanilla-SA group MLP: one MFMA contraction per call (csrc/sa_mlp.hip, lin_cols_kernel) -----
def linear_cols_pack(w, n_out, k, transposed_source=False, gather_order=False):
    """Packed copy of a weight matrix for linear_cols / sa_gather_linear (A-fragment order of the f32 MFMA)."""
    lib = _lib.load()
    wf = torch.empty((_cols_packed_size(int(n_out), int(k)),), dtype=F32, device=w.device)
    _numel_ok(w, n_out * k, "w")
    _call("pda_linear_cols_pack", w, _chk(w, "w", F32), _chk(wf, "wf", F32), n_out, k, 1 if transposed_source else 0,
          1 if gather_order else 0)
    return wf


def linear_cols(x, wf, y, tokens, k, n_out):
    _numel_ok(x, tokens * k, "x"); _numel_ok(y, tokens * n_out, "y")
    _numel_ok(wf, _cols_packed_size(int(n_out), int(k)), "wf")
    _call("pda_linear_cols", x, _chk(x, "x", F32), _chk(wf, "wf", F32), _chk(y, "y", F32), tokens, k, n_out)
    return 1


# ---- f32 GEMM on the bf16 matrix cores, operands split into three bf16 terms (csrc/gemm_split.hip) -----------------
def linear_split_pack(w, n_out, k, transposed_source=False):
    lib = _lib.load()
    wf = torch.empty((_split_packed_bytes(int(n_out), int(k)),), dtype=torch.uint8, device=w.device)
    _numel_ok(w, n_out * k, "w")
    _call("pda_linear_split_pack", w, _chk(w, "w", F32), _chk(wf, "wf", torch.uint8), n_out, k, 1 if transposed_source else 0)
    return wf


def linear_split_pack_both(w, n_out, k):
    """(planes of W, planes of W^T) of one weight W (n_out, k) in one launch (csrc/gemm_split.hip)."""
    wf = torch.empty((_split_packed_bytes(int(n_out), int(k)),), dtype=torch.uint8, device=w.device)
    wft = torch.empty((_split_packed_bytes(int(k), int(n_out)),), dtype=torch.uint8, device=w.device)
    _numel_ok(w, n_out * k, "w")
    _call("pda_linear_split_pack_both", w, _chk(w, "w", F32), _chk(wf, "wf", torch.uint8), _chk(wft, "wft", torch.uint8), n_out, k)
    return wf, wft


def linear_split(x, wf, bias, y, tokens, k, n_out, relu=False):
    _numel_ok(x, tokens * k, "x"); _numel_ok(y, tokens * n_out, "y")
    _numel_ok(wf, _split_packed_bytes(int(n_out), int(k)), "wf")
    if bias is not None:
        _numel_ok(bias, n_out, "bias")
    _call("pda_linear_split", x, _chk(x, "x", F32), _chk(wf, "wf", torch.uint8), None if bias is None else _chk(bias, "bias", F32),
          _chk(y, "y", F32), tokens, k, n_out, 1 if relu else 0)
    return 1


def gemm_split(x, wf, bias, y, tokens, k, n_out, relu=False, accumulate=False):
    """y (tokens, n_out) [+]= x (tokens, k) W^T [+ bias] [relu] on gemm_split_kernel (any k % 32 == 0, any n_out)."""
    _numel_ok(x, tokens * k, "x"); _numel_ok(y, tokens * n_out, "y")
    _numel_ok(wf, _split_packed_bytes(int(n_out), int(k)), "wf")
    if bias is not None:
        _numel_ok(bias, n_out, "bias")
    _call("pda_gemm_split", x, _chk(x, "x", F32), _chk(wf, "wf", torch.uint8), None if bias is None else _chk(bias, "bias", F32),
          _chk(y, "y", F32), tokens, k, n_out, 1 if relu else 0, 1 if accumulate else 0)
    return 1


def sa_gather_linear(xyz, new_xyz, feats_pm, idx, wf, y, b, n, m, c, nsample, n_out):
    _numel_ok(xyz, b * n * 3, "xyz"); _numel_ok(new_xyz, b * m * 3, "new_xyz"); _numel_ok(feats_pm, b * n * c, "feats_pm")
    _numel_ok(idx, b * m * nsample, "idx"); _numel_ok(y, b * m * nsample * n_out, "y")
    _numel_ok(wf, _cols_packed_size(int(n_out), 3 + int(c)), "wf")
    _call("pda_sa_gather_linear", xyz, _chk(xyz, "xyz", F32), _chk(new_xyz, "new_xyz", F32), _chk(feats_pm, "feats_pm", F32),
          _chk(idx, "idx", I32), _chk(wf, "wf", F32), _chk(y, "y", F32), b, n, m, c, nsample, n_out)
    return 1


# ---- the narrow vanilla SA scale in training form (csrc/sa_train_small.hip) -------------------------------------------
@functools.lru_cache(maxsize=None)
def sa_small_train_workspace_bytes():
    return int(_lib.load().pda_sa_small_train_workspace_bytes())


@functools.lru_cache(maxsize=None)
def sa_small_train_supported(c, ns, c1, c2, c3, tokens):
    return bool(_lib.load().pda_sa_small_train_supported(int(c), int(ns), int(c1), int(c2), int(c3), int(tokens)))


def _ptr_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def sa_small_train_fwd(xyz, new_xyz, feat_pm, idx, weights, gammas, betas, running_means, running_vars, eps, momentum,
                       workspace, out, zmax, arg, b, n, m, c, ns):
    """MI355X extension: forward of [group -> (conv1x1 -> BN(batch stats) -> ReLU) x 3 -> max] for the narrow chains of SA
    layer 0 as recompute passes; out / zmax (b*m, c3) float, arg (b*m, c3) uint8.  weights = (w1, w2, w3) as (c_out, c_in)."""
    c1, c2, c3 = (int(w.shape[0]) for w in weights)
    tokens = b * m * ns
    _numel_ok(xyz, b * n * 3, "xyz"); _numel_ok(new_xyz, b * m * 3, "new_xyz"); _numel_ok(idx, tokens, "idx")
    _numel_ok(out, b * m * c3, "out"); _numel_ok(zmax, b * m * c3, "zmax"); _numel_ok(arg, b * m * c3, "arg")
    _numel_ok(workspace, sa_small_train_workspace_bytes(), "workspace")
    assert weights[0].shape[1] == 3 + c and weights[1].shape[1] == c1 and weights[2].shape[1] == c2
    if feat_pm is not None:
        _numel_ok(feat_pm, b * n * c, "feat_pm")
    rm = [None if t is None else _chk(t, "running_mean", F32) for t in running_means]
    rv = [None if t is None else _chk(t, "running_var", F32) for t in running_vars]
    _buffers_written(rm[0] or rm[1] or rm[2])
    _call("pda_sa_small_train_fwd", xyz, _chk(xyz, "xyz", F32), _chk(new_xyz, "new_xyz", F32),
          None if feat_pm is None else _chk(feat_pm, "feat_pm", F32), _chk(idx, "idx", I32),
          _chk(weights[0], "w1", F32), _chk(weights[1], "w2", F32), _chk(weights[2], "w3", F32),
          _ptr_array([_chk(t, "gamma", F32) for t in gammas]), _ptr_array([_chk(t, "beta", F32) for t in betas]),
          _ptr_array(rm), _ptr_array(rv), (ctypes.c_float * 3)(*[float(e) for e in eps]),
          (ctypes.c_float * 3)(*[float(x) for x in momentum]), _chk(workspace, "workspace", torch.uint8),
          _chk(out, "out", F32), _chk(zmax, "zmax", F32), _chk(arg, "arg", torch.uint8), b, n, m, c, ns, c1, c2, c3)
    return 1


def sa_small_train_bwd(xyz, new_xyz, feat_pm, idx, grad_out, zmax, arg, workspace, dz2, dz1, dws, dgammas, dbetas, b, n, m, c, ns):
    c1, c2, c3 = (int(w.shape[0]) for w in dws)
    tokens = b * m * ns
    _numel_ok(grad_out, b * m * c3, "grad_out"); _numel_ok(dz2, tokens * c2, "dz2"); _numel_ok(dz1, tokens * c1, "dz1")
    _numel_ok(workspace, sa_small_train_workspace_bytes(), "workspace")
    _call("pda_sa_small_train_bwd", xyz, _chk(xyz, "xyz", F32), _chk(new_xyz, "new_xyz", F32),
          None if feat_pm is None else _chk(feat_pm, "feat_pm", F32), _chk(idx, "idx", I32), _chk(grad_out, "grad_out", F32),
          _chk(zmax, "zmax", F32), _chk(arg, "arg", torch.uint8), _chk(workspace, "workspace", torch.uint8),
          _chk(dz2, "dz2", F32), _chk(dz1, "dz1", F32), _chk(dws[0], "dw1", F32), _chk(dws[1], "dw2", F32), _chk(dws[2], "dw3", F32),
          _ptr_array([_chk(t, "dgamma", F32) for t in dgammas]), _ptr_array([_chk(t, "dbeta", F32) for t in dbetas]),
          b, n, m, c, ns, c1, c2, c3)
    return 1


@functools.lru_cache(maxsize=None)
def sa_xyz_grad_scratch_bytes(c1):
    return int(_lib.load().pda_sa_xyz_grad_scratch_bytes(int(c1)))


def sa_xyz_grad(grad_z1, xyz, new_xyz, idx, w, dw, grad_new_xyz, b, n, m, ns, c1):
    """MI355X extension: the 3 coordinate columns of a vanilla SA scale's first layer in the backward pass
    (csrc/sa_xyz_grad.hip): dw[:, 0:3] and grad_new_xyz (b, m, 3) from grad_z1 (b*m*ns, c1); w / dw are (c1, 3 + C)."""
    _numel_ok(grad_z1, b * m * ns * c1, "grad_z1"); _numel_ok(idx, b * m * ns, "idx")
    _numel_ok(xyz, b * n * 3, "xyz"); _numel_ok(new_xyz, b * m * 3, "new_xyz")
    assert w.shape[0] == c1 and dw.shape == w.shape and w.shape[1] >= 3
    scratch = torch.empty((sa_xyz_grad_scratch_bytes(c1),), dtype=torch.uint8, device=xyz.device)
    _call("pda_sa_xyz_grad", xyz, _chk(grad_z1, "grad_z1", F32), _chk(xyz, "xyz", F32), _chk(new_xyz, "new_xyz", F32),
          _chk(idx, "idx", I32), _chk(w, "w", F32), int(w.shape[1]), _chk(dw, "dw", F32), int(dw.shape[1]),
          None if grad_new_xyz is None else _chk(grad_new_xyz, "grad_new_xyz", F32), _chk(scratch, "scratch", torch.uint8),
          b, n, m, ns, c1)
    return 1


def sa_point_gather(point_rows, xyz, new_xyz, idx, w, z, b, n, m, ns, c1, bias=None, relu=False):
    """MI355X extension: z (b*m*ns, c1) = point_rows[idx] + w[:, 0:3] (xyz[idx] - new_xyz[centre]) (csrc/sa_xyz_grad.hip): the
    first layer of a wide SA scale from its per-point projection point_rows (b*n, c1) = features w[:, 3:]^T."""
    _numel_ok(point_rows, b * n * c1, "point_rows"); _numel_ok(z, b * m * ns * c1, "z"); _numel_ok(idx, b * m * ns, "idx")
    _numel_ok(xyz, b * n * 3, "xyz"); _numel_ok(new_xyz, b * m * 3, "new_xyz")
    assert w.shape[0] == c1 and w.shape[1] >= 3
    _call("pda_sa_point_gather", xyz, _chk(point_rows, "point_rows", F32), _chk(xyz, "xyz", F32), _chk(new_xyz, "new_xyz", F32),
          _chk(idx, "idx", I32), _chk(w, "w", F32), int(w.shape[1]), None if bias is None else _chk(bias, "bias", F32),
          1 if relu else 0, _chk(z, "z", F32), b, n, m, ns, c1)
    return 1
