// group_gather.hip -- gather / group (+ their gradients) for gfx950.
// Reference kernels: gather_points{,_grad}_kernel_fast (sampling_gpu.cu:8-24, :46-63) and
// group_points{,_grad}_kernel_fast (group_points_gpu.cu:53-72, :14-31)
// under /root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/.
//
// The reference launches one thread per (channel, element) and re-reads idx once per channel.
// Here a thread owns VEC consecutive output elements: it loads their indices ONCE (one 16-B
// load), then walks a chunk of channels, so every store is a coalesced 16-B store and idx
// traffic drops by C.  The random 4-byte reads of `points` stay (the (B,C,N) layout is the
// boundary's), but a channel row is N*4 B (64 KiB at N=16384) and is served from L2.
// Gradients use no-return global_atomic_add_f32, like the reference's atomicAdd, so the
// summation order (and the last bits) are run-to-run non-deterministic, as in the reference.
#include "pda_common.h"

namespace pda {

constexpr int GG_THREADS = 256;
constexpr int GG_VEC = 4;
constexpr int GG_CCHUNK = 8;  // channels per workgroup along grid.y

// out[b,c,e] = points[b,c,idx[b,e]]  for e in [0,E): covers gather (E=m) and group (E=npoints*nsample)
__global__ __launch_bounds__(GG_THREADS) void gather_rows_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ idx, float* __restrict__ out,
    int c, int n, int64_t E) {
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * GG_CCHUNK;
    const int c1 = min(c, c0 + GG_CCHUNK);
    const int64_t e0 = ((int64_t)blockIdx.x * GG_THREADS + threadIdx.x) * GG_VEC;
    if (e0 >= E) return;
    const int32_t* id = idx + (int64_t)bs * E + e0;
    const float* src = points + ((int64_t)bs * c + c0) * n;
    float* dst = out + ((int64_t)bs * c + c0) * E + e0;
    const bool vec_ok = e0 + GG_VEC <= E && (E % GG_VEC) == 0 &&
                        ((((uintptr_t)id) | ((uintptr_t)dst)) & 15) == 0;
    if (vec_ok) {
        const int4 k = *reinterpret_cast<const int4*>(id);
        for (int ci = c0; ci < c1; ++ci, src += n, dst += E) {
            float4 v;
            v.x = src[k.x]; v.y = src[k.y]; v.z = src[k.z]; v.w = src[k.w];
            *reinterpret_cast<float4*>(dst) = v;
        }
    } else {
        const int cntv = (int)min((int64_t)GG_VEC, E - e0);
        for (int ci = c0; ci < c1; ++ci, src += n, dst += E)
            for (int v = 0; v < cntv; ++v) dst[v] = src[id[v]];
    }
}

// grad_points[b,c,idx[b,e]] += grad_out[b,c,e]
__global__ __launch_bounds__(GG_THREADS) void scatter_add_rows_kernel(
    const float* __restrict__ grad_out, const int32_t* __restrict__ idx,
    float* __restrict__ grad_points, int c, int n, int64_t E) {
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * GG_CCHUNK;
    const int c1 = min(c, c0 + GG_CCHUNK);
    const int64_t e0 = ((int64_t)blockIdx.x * GG_THREADS + threadIdx.x) * GG_VEC;
    if (e0 >= E) return;
    const int32_t* id = idx + (int64_t)bs * E + e0;
    const float* src = grad_out + ((int64_t)bs * c + c0) * E + e0;
    float* dst = grad_points + ((int64_t)bs * c + c0) * n;
    const bool vec_ok = e0 + GG_VEC <= E && (E % GG_VEC) == 0 &&
                        ((((uintptr_t)id) | ((uintptr_t)src)) & 15) == 0;
    if (vec_ok) {
        const int4 k = *reinterpret_cast<const int4*>(id);
        for (int ci = c0; ci < c1; ++ci, src += E, dst += n) {
            const float4 g = *reinterpret_cast<const float4*>(src);
            atomicAdd(dst + k.x, g.x);
            atomicAdd(dst + k.y, g.y);
            atomicAdd(dst + k.z, g.z);
            atomicAdd(dst + k.w, g.w);
        }
    } else {
        const int cntv = (int)min((int64_t)GG_VEC, E - e0);
        for (int ci = c0; ci < c1; ++ci, src += E, dst += n)
            for (int v = 0; v < cntv; ++v) atomicAdd(dst + id[v], src[v]);
    }
}

static int launch_rows(bool grad, const float* a, const int32_t* idx, float* o, int b, int c, int n,
                       int64_t E, hipStream_t stream, const char* what) {
    PDA_REQUIRE(b >= 0 && c >= 0 && n >= 0 && E >= 0, "%s: negative size", what);
    if (b == 0 || c == 0 || E == 0) return PDA_OK;
    PDA_REQUIRE(a && idx && o, "%s: null pointer", what);
    PDA_REQUIRE(n > 0, "%s: n == 0 with a non-empty index list", what);
    PDA_REQUIRE(b <= 65535 && divup(c, GG_CCHUNK) <= 65535, "%s: b or c too large for the grid", what);
    const int64_t gx = divup64(E, (int64_t)GG_THREADS * GG_VEC);
    PDA_REQUIRE(gx < INT32_MAX, "%s: too many elements", what);
    dim3 grid((unsigned)gx, divup(c, GG_CCHUNK), b), block(GG_THREADS);
    if (grad) hipLaunchKernelGGL(scatter_add_rows_kernel, grid, block, 0, stream, a, idx, o, c, n, E);
    else hipLaunchKernelGGL(gather_rows_kernel, grid, block, 0, stream, a, idx, o, c, n, E);
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_gather_points(const float* points, const int32_t* idx, float* out, int b, int c, int n,
                              int m, pda_stream_t stream) {
    return pda::launch_rows(false, points, idx, out, b, c, n, m, (hipStream_t)stream, "pda_gather_points");
}

PDA_API int pda_gather_points_grad(const float* grad_out, const int32_t* idx, float* grad_points, int b,
                                   int c, int n, int m, pda_stream_t stream) {
    return pda::launch_rows(true, grad_out, idx, grad_points, b, c, n, m, (hipStream_t)stream,
                            "pda_gather_points_grad");
}

PDA_API int pda_group_points(const float* points, const int32_t* idx, float* out, int b, int c, int n,
                             int npoints, int nsample, pda_stream_t stream) {
    if (npoints < 0 || nsample < 0) {
        pda::set_error("pda_group_points: negative size");
        return PDA_ERR_INVALID_ARGUMENT;
    }
    return pda::launch_rows(false, points, idx, out, b, c, n, (int64_t)npoints * nsample,
                            (hipStream_t)stream, "pda_group_points");
}

PDA_API int pda_group_points_grad(const float* grad_out, const int32_t* idx, float* grad_points, int b,
                                  int c, int n, int npoints, int nsample, pda_stream_t stream) {
    if (npoints < 0 || nsample < 0) {
        pda::set_error("pda_group_points_grad: negative size");
        return PDA_ERR_INVALID_ARGUMENT;
    }
    return pda::launch_rows(true, grad_out, idx, grad_points, b, c, n, (int64_t)npoints * nsample,
                            (hipStream_t)stream, "pda_group_points_grad");
}

// ---- point-major ("rows") gather: the layout the MI355X pipeline uses internally ------------
// out[b,e,:] = rows[b, idx[b,e], :]   rows (B,N,C), idx (B,E), out (B,E,C).
// A neighbour's feature vector is ONE contiguous C*4-byte row, so the gather is a coalesced row
// copy (16 B per lane) instead of the C scattered 4-byte reads of the channel-major
// group_points layout, and the gradient is a scatter-add of contiguous rows (the float-atomic
// shape that runs at full rate, MI355X_MICROARCH.md "Global float atomics").
// No reference counterpart: the reference only has the (B,C,N) layout (group_points_gpu.cu).
namespace pda {

template <bool GRAD>
__global__ __launch_bounds__(256) void group_rows_kernel(const float* __restrict__ src,
                                                          const int32_t* __restrict__ idx,
                                                          float* __restrict__ dst, int n, int c, int64_t E) {
    const int bs = blockIdx.y;
    const int c4 = c >> 2;                                  // float4 per row (c % 4 == 0 path)
    const int64_t total = E * c4;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int64_t e = o / c4;
    const int q = (int)(o - e * c4);
    const int k = idx[(int64_t)bs * E + e];
    if (!GRAD) {
        const float4 v = reinterpret_cast<const float4*>(src + ((int64_t)bs * n + k) * c)[q];
        reinterpret_cast<float4*>(dst + ((int64_t)bs * E + e) * c)[q] = v;
    } else {
        const float4 g = reinterpret_cast<const float4*>(src + ((int64_t)bs * E + e) * c)[q];
        float* d = dst + ((int64_t)bs * n + k) * c + 4 * q;
        atomicAdd(d + 0, g.x); atomicAdd(d + 1, g.y); atomicAdd(d + 2, g.z); atomicAdd(d + 3, g.w);
    }
}

template <bool GRAD>
__global__ __launch_bounds__(256) void group_rows_scalar_kernel(const float* __restrict__ src,
                                                                 const int32_t* __restrict__ idx,
                                                                 float* __restrict__ dst, int n, int c, int64_t E) {
    const int bs = blockIdx.y;
    const int64_t total = E * c;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int64_t e = o / c;
    const int q = (int)(o - e * c);
    const int k = idx[(int64_t)bs * E + e];
    if (!GRAD) dst[((int64_t)bs * E + e) * c + q] = src[((int64_t)bs * n + k) * c + q];
    else atomicAdd(dst + ((int64_t)bs * n + k) * c + q, src[((int64_t)bs * E + e) * c + q]);
}

static int launch_group_rows(bool grad, const float* src, const int32_t* idx, float* dst, int b, int n, int c,
                             int64_t E, hipStream_t stream, const char* what) {
    PDA_REQUIRE(b >= 0 && n >= 0 && c >= 0 && E >= 0, "%s: negative size", what);
    if (b == 0 || c == 0 || E == 0) return PDA_OK;
    PDA_REQUIRE(src && idx && dst && n > 0, "%s: null pointer or n == 0", what);
    PDA_REQUIRE(b <= 65535, "%s: b too large", what);
    // gradients: one float per lane, so a wave-instruction adds 256 CONTIGUOUS bytes of one or two
    // rows -- the shape float atomics run at full rate in (MI355X_MICROARCH.md); the float4 mapping
    // spreads one atomic instruction over 16-byte-strided words of 4 rows and measured 4x slower
    const bool vec = !grad && (c % 4 == 0) && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    const int64_t total = vec ? E * (c / 4) : E * c;
    const int64_t gx = divup64(total, 256);
    PDA_REQUIRE(gx < INT32_MAX, "%s: too many elements", what);
    dim3 grid((unsigned)gx, b), block(256);
    if (vec) {
        if (grad) hipLaunchKernelGGL(group_rows_kernel<true>, grid, block, 0, stream, src, idx, dst, n, c, E);
        else hipLaunchKernelGGL(group_rows_kernel<false>, grid, block, 0, stream, src, idx, dst, n, c, E);
    } else {
        if (grad) hipLaunchKernelGGL(group_rows_scalar_kernel<true>, grid, block, 0, stream, src, idx, dst, n, c, E);
        else hipLaunchKernelGGL(group_rows_scalar_kernel<false>, grid, block, 0, stream, src, idx, dst, n, c, E);
    }
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_group_rows(const float* rows, const int32_t* idx, float* out, int b, int n, int c,
                           int64_t num_idx, pda_stream_t stream) {
    return pda::launch_group_rows(false, rows, idx, out, b, n, c, num_idx, (hipStream_t)stream, "pda_group_rows");
}

PDA_API int pda_group_rows_grad(const float* grad_out, const int32_t* idx, float* grad_rows, int b, int n,
                                int c, int64_t num_idx, pda_stream_t stream) {
    return pda::launch_group_rows(true, grad_out, idx, grad_rows, b, n, c, num_idx, (hipStream_t)stream,
                                  "pda_group_rows_grad");
}
