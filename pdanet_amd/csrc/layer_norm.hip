// layer_norm.hip -- LayerNorm over the last dimension with an optional fused residual add, forward and
// backward (include/pda_train.h).  The PDA layer's TransformerEncoderLayerPreNorm (PointFormer.py:28-38)
// normalises the (tokens, D) activations twice per scale; D = 256 / 512, tokens = 131k-262k.  Through torch
// that is a forward kernel plus TWO backward kernels that each re-read x and dy (cuComputeGradInput,
// cuComputePartGradGammaBeta).  Here: one wave per row, the row lives in registers (D/64 floats per lane,
// 16-byte accesses), mean / variance by DPP-free butterfly shuffles; backward is ONE pass: dx is written and
// every lane keeps running sums of dy and dy*xhat for its own columns, reduced over the block in LDS and
// written as per-block partials that a small second kernel adds in fixed order (deterministic).
#include "pda_common.h"

namespace pda {

constexpr int LN_BLOCKS = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// V = float4 chunks per lane: D = 256 * V.  TX: element type of x (float, or bf16 when x comes out of a bf16 GEMM);
// y (fp32) and yb (a bf16 copy for the next bf16 GEMM) are each optional.
template <int V, bool RESIDUAL, typename TX>
__global__ __launch_bounds__(256) void layer_norm_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ sum_out, float* __restrict__ y,
                                                             bf16_t* __restrict__ yb, float* __restrict__ mean_rstd, int64_t rows,
                                                             float eps) {
    constexpr int D = 256 * V;
    const int lane = lane_id();
    float4 g[V], be[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        g[v] = reinterpret_cast<const float4*>(gamma)[v * 64 + lane];
        be[v] = reinterpret_cast<const float4*>(beta)[v * 64 + lane];
    }
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave_id(); r < rows; r += (int64_t)gridDim.x * 4) {
        float4 a[V];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            a[v] = load4(x + r * D + 4 * (v * 64 + lane));
            if (RESIDUAL) {
                const float4 b = reinterpret_cast<const float4*>(res + r * D)[v * 64 + lane];
                a[v].x += b.x; a[v].y += b.y; a[v].z += b.z; a[v].w += b.w;
                reinterpret_cast<float4*>(sum_out + r * D)[v * 64 + lane] = a[v];
            }
            s += (a[v].x + a[v].y) + (a[v].z + a[v].w);
        }
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float dx = a[v].x - mean, dy = a[v].y - mean, dz = a[v].z - mean, dw = a[v].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float4 o;
            o.x = (a[v].x - mean) * rstd * g[v].x + be[v].x;
            o.y = (a[v].y - mean) * rstd * g[v].y + be[v].y;
            o.z = (a[v].z - mean) * rstd * g[v].z + be[v].z;
            o.w = (a[v].w - mean) * rstd * g[v].w + be[v].w;
            if (y) reinterpret_cast<float4*>(y + r * D)[v * 64 + lane] = o;
            if (yb) store4(yb + r * D + 4 * (v * 64 + lane), o);
        }
        if (lane == 0) { mean_rstd[r * 2] = mean; mean_rstd[r * 2 + 1] = rstd; }
    }
}

// TWO: the incoming gradient is dy + dy2 (a residual branch and a projection's input gradient), added on the fly
// (T2: element type of dy2; dxb: optional bf16 copy of dx for the bf16 GEMMs that consume it)
template <int V, bool TWO, typename T2>
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const T2* __restrict__ dy2, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean_rstd, float* __restrict__ dx,
                                                             bf16_t* __restrict__ dxb, float* __restrict__ partial, int64_t rows) {
    constexpr int D = 256 * V;
    __shared__ float red[2][4][D];
    const int lane = lane_id(), w = wave_id();
    float4 g[V], sb[V], sg[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        g[v] = reinterpret_cast<const float4*>(gamma)[v * 64 + lane];
        sb[v] = make_float4(0, 0, 0, 0); sg[v] = make_float4(0, 0, 0, 0);
    }
    for (int64_t r = (int64_t)blockIdx.x * 4 + w; r < rows; r += (int64_t)gridDim.x * 4) {
        const float mean = mean_rstd[r * 2], rstd = mean_rstd[r * 2 + 1];
        float4 xh[V], d[V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float4 xv = reinterpret_cast<const float4*>(x + r * D)[v * 64 + lane];
            d[v] = reinterpret_cast<const float4*>(dy + r * D)[v * 64 + lane];
            if (TWO) {
                const float4 e = load4(dy2 + r * D + 4 * (v * 64 + lane));
                d[v].x += e.x; d[v].y += e.y; d[v].z += e.z; d[v].w += e.w;
            }
            xh[v] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            sb[v].x += d[v].x; sb[v].y += d[v].y; sb[v].z += d[v].z; sb[v].w += d[v].w;
            sg[v].x += d[v].x * xh[v].x; sg[v].y += d[v].y * xh[v].y; sg[v].z += d[v].z * xh[v].z; sg[v].w += d[v].w * xh[v].w;
            d[v].x *= g[v].x; d[v].y *= g[v].y; d[v].z *= g[v].z; d[v].w *= g[v].w;        // dy * gamma
            s1 += (d[v].x + d[v].y) + (d[v].z + d[v].w);
            s2 += (d[v].x * xh[v].x + d[v].y * xh[v].y) + (d[v].z * xh[v].z + d[v].w * xh[v].w);
        }
        const float m1 = wave_sum(s1) * (1.0f / D), m2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float4 o;
            o.x = rstd * (d[v].x - m1 - xh[v].x * m2);
            o.y = rstd * (d[v].y - m1 - xh[v].y * m2);
            o.z = rstd * (d[v].z - m1 - xh[v].z * m2);
            o.w = rstd * (d[v].w - m1 - xh[v].w * m2);
            reinterpret_cast<float4*>(dx + r * D)[v * 64 + lane] = o;
            if (dxb) store4(dxb + r * D + 4 * (v * 64 + lane), o);
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        reinterpret_cast<float4*>(red[0][w])[v * 64 + lane] = sb[v];
        reinterpret_cast<float4*>(red[1][w])[v * 64 + lane] = sg[v];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        partial[((size_t)blockIdx.x * 2 + 0) * D + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        partial[((size_t)blockIdx.x * 2 + 1) * D + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

// dbeta / dgamma = sum of the per-block partials; 64 columns x 16 slices per block of 1024 threads
__global__ __launch_bounds__(1024) void layer_norm_finalize_kernel(const float* __restrict__ partial, int nblocks, int d,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[2][16][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f, b = 0.f;
    if (c < d) {
        float va[8], vb[8];
        for (int k0 = part; k0 < nblocks; k0 += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 16 * u;
                va[u] = k < nblocks ? partial[((size_t)k * 2 + 0) * d + c] : 0.f;
                vb[u] = k < nblocks ? partial[((size_t)k * 2 + 1) * d + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += va[u]; b += vb[u]; }
        }
    }
    red[0][part][cl] = a; red[1][part][cl] = b;
    __syncthreads();
    if (part != 0 || c >= d) return;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) { s1 += red[0][p][cl]; s2 += red[1][p][cl]; }
    dbeta[c] = s1;
    dgamma[c] = s2;
}

static int ln_grid(int64_t rows) {
    const int64_t b = divup64(rows, 4);
    return (int)(b < LN_BLOCKS ? b : LN_BLOCKS);
}

}  // namespace pda

PDA_API int64_t pda_layer_norm_scratch_bytes(int d) { return (int64_t)pda::LN_BLOCKS * 2 * (d > 0 ? d : 0) * (int64_t)sizeof(float); }

namespace pda {

template <typename TX>
static int launch_layer_norm_fwd(const TX* x, const float* residual, const float* gamma, const float* beta, float* sum_out, float* y,
                                 bf16_t* yb, float* mean_rstd, int64_t rows, int d, float eps, hipStream_t st, const char* what) {
    PDA_REQUIRE(rows >= 0, "%s: rows = %lld", what, (long long)rows);
    PDA_REQUIRE(d == 256 || d == 512 || d == 1024, "%s: D = %d (256, 512 or 1024)", what, d);
    if (rows == 0) return PDA_OK;
    PDA_REQUIRE(x && gamma && beta && (y || yb) && mean_rstd, "%s: null pointer", what);
    PDA_REQUIRE((residual == nullptr) == (sum_out == nullptr), "%s: residual and sum_out come together", what);
    PDA_REQUIRE((((uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)residual | (uintptr_t)sum_out) & 15) == 0 &&
                    (((uintptr_t)x | (uintptr_t)yb) & (4 * sizeof(TX) - 1) & 15) == 0 && ((uintptr_t)yb & 7) == 0,
                "%s: pointers must be 16-byte aligned (8 for bf16 tensors)", what);
    const dim3 grid(ln_grid(rows)), block(256);
#define PDA_LN_FWD(V)                                                                                                          \
    if (residual) hipLaunchKernelGGL((layer_norm_fwd_kernel<V, true, TX>), grid, block, 0, st, x, residual, gamma, beta, sum_out, y, yb, mean_rstd, rows, eps); \
    else hipLaunchKernelGGL((layer_norm_fwd_kernel<V, false, TX>), grid, block, 0, st, x, residual, gamma, beta, sum_out, y, yb, mean_rstd, rows, eps)
    if (d == 256) { PDA_LN_FWD(1); } else if (d == 512) { PDA_LN_FWD(2); } else { PDA_LN_FWD(4); }
#undef PDA_LN_FWD
    return check_launch(what);
}

template <typename T2>
static int launch_layer_norm_bwd(const float* x, const float* grad_y, const T2* grad_y2, const float* gamma, const float* mean_rstd,
                                 float* grad_x, bf16_t* grad_x_b, float* grad_gamma, float* grad_beta, void* scratch, int64_t rows, int d,
                                 hipStream_t st, const char* what) {
    PDA_REQUIRE(rows >= 1, "%s: rows = %lld", what, (long long)rows);
    PDA_REQUIRE(d == 256 || d == 512 || d == 1024, "%s: D = %d (256, 512 or 1024)", what, d);
    PDA_REQUIRE(x && grad_y && gamma && mean_rstd && grad_x && grad_gamma && grad_beta && scratch, "%s: null pointer", what);
    PDA_REQUIRE((((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_x | (uintptr_t)gamma) & 15) == 0 &&
                    ((uintptr_t)grad_y2 & (4 * sizeof(T2) - 1)) == 0 && ((uintptr_t)grad_x_b & 7) == 0,
                "%s: pointers must be 16-byte aligned (8 for bf16 tensors)", what);
    const int nblocks = ln_grid(rows);
    const dim3 grid(nblocks), block(256);
    float* partial = (float*)scratch;
#define PDA_LN_BWD(V)                                                                                                               \
    if (grad_y2) hipLaunchKernelGGL((layer_norm_bwd_kernel<V, true, T2>), grid, block, 0, st, x, grad_y, grad_y2, gamma, mean_rstd, grad_x, grad_x_b, partial, rows); \
    else hipLaunchKernelGGL((layer_norm_bwd_kernel<V, false, T2>), grid, block, 0, st, x, grad_y, grad_y2, gamma, mean_rstd, grad_x, grad_x_b, partial, rows)
    if (d == 256) { PDA_LN_BWD(1); } else if (d == 512) { PDA_LN_BWD(2); } else { PDA_LN_BWD(4); }
#undef PDA_LN_BWD
    hipLaunchKernelGGL(layer_norm_finalize_kernel, dim3(divup(d, 64)), dim3(1024), 0, st, partial, nblocks, d, grad_gamma, grad_beta);
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_layer_norm_fwd(const float* x, const float* residual, const float* gamma, const float* beta, float* sum_out,
                               float* y, float* mean_rstd, int64_t rows, int d, float eps, pda_stream_t stream) {
    PDA_REQUIRE(y != nullptr || rows == 0, "pda_layer_norm_fwd: null pointer");
    return pda::launch_layer_norm_fwd<float>(x, residual, gamma, beta, sum_out, y, nullptr, mean_rstd, rows, d, eps, (hipStream_t)stream,
                                             "pda_layer_norm_fwd");
}

PDA_API int pda_layer_norm_fwd_mixed(const void* x, int x_is_bf16, const float* residual, const float* gamma, const float* beta,
                                     float* sum_out, float* y, uint16_t* y_bf16, float* mean_rstd, int64_t rows, int d, float eps,
                                     pda_stream_t stream) {
    if (x_is_bf16)
        return pda::launch_layer_norm_fwd<pda::bf16_t>((const pda::bf16_t*)x, residual, gamma, beta, sum_out, y, y_bf16, mean_rstd, rows, d,
                                                       eps, (hipStream_t)stream, "pda_layer_norm_fwd_mixed");
    return pda::launch_layer_norm_fwd<float>((const float*)x, residual, gamma, beta, sum_out, y, y_bf16, mean_rstd, rows, d, eps,
                                             (hipStream_t)stream, "pda_layer_norm_fwd_mixed");
}

PDA_API int pda_layer_norm_bwd(const float* x, const float* grad_y, const float* grad_y2, const float* gamma, const float* mean_rstd,
                               float* grad_x, float* grad_gamma, float* grad_beta, void* scratch, int64_t rows, int d,
                               pda_stream_t stream) {
    return pda::launch_layer_norm_bwd<float>(x, grad_y, grad_y2, gamma, mean_rstd, grad_x, nullptr, grad_gamma, grad_beta, scratch, rows, d,
                                             (hipStream_t)stream, "pda_layer_norm_bwd");
}

PDA_API int pda_layer_norm_bwd_mixed(const float* x, const float* grad_y, const void* grad_y2, int grad_y2_is_bf16, const float* gamma,
                                     const float* mean_rstd, float* grad_x, uint16_t* grad_x_bf16, float* grad_gamma, float* grad_beta,
                                     void* scratch, int64_t rows, int d, pda_stream_t stream) {
    if (grad_y2_is_bf16)
        return pda::launch_layer_norm_bwd<pda::bf16_t>(x, grad_y, (const pda::bf16_t*)grad_y2, gamma, mean_rstd, grad_x, grad_x_bf16,
                                                       grad_gamma, grad_beta, scratch, rows, d, (hipStream_t)stream, "pda_layer_norm_bwd_mixed");
    return pda::launch_layer_norm_bwd<float>(x, grad_y, (const float*)grad_y2, gamma, mean_rstd, grad_x, grad_x_bf16, grad_gamma, grad_beta,
                                             scratch, rows, d, (hipStream_t)stream, "pda_layer_norm_bwd_mixed");
}

// ---- residual add + max-pool over the tokens of a group -------------------------------------------------
// The tail of a PDA scale: y = src + ffn (T x D) followed by the max over the nsample tokens of each group
// (pointnet2_modules.py:929-931).  Fused: y is never written; out (G, D) and the arg-max token (first maximum)
// for the backward pass, which writes the dense (G, S, D) gradient in one pass (zeros + the routed values).
namespace pda {

template <typename TB>
__global__ __launch_bounds__(256) void add_max_pool_kernel(const float* __restrict__ a, const TB* __restrict__ b,
                                                           float* __restrict__ out, uint8_t* __restrict__ arg, int64_t groups,
                                                           int s, int d4) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (group, 4-channel column)
    if (e >= groups * d4) return;
    const int64_t g = e / d4;
    const int c = (int)(e % d4);
    const float4* pa = reinterpret_cast<const float4*>(a) + (size_t)g * s * d4 + c;
    const TB* pb = b + 4 * ((size_t)g * s * d4 + c);
    float4 best = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
    uchar4 bi = make_uchar4(0, 0, 0, 0);
    for (int t = 0; t < s; ++t) {
        const float4 x = pa[(size_t)t * d4], y = load4(pb + 4 * (size_t)t * d4);
        const float4 v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        if (v.x > best.x) { best.x = v.x; bi.x = (uint8_t)t; }
        if (v.y > best.y) { best.y = v.y; bi.y = (uint8_t)t; }
        if (v.z > best.z) { best.z = v.z; bi.z = (uint8_t)t; }
        if (v.w > best.w) { best.w = v.w; bi.w = (uint8_t)t; }
    }
    reinterpret_cast<float4*>(out)[e] = best;
    reinterpret_cast<uchar4*>(arg)[e] = bi;
}

__global__ __launch_bounds__(256) void max_pool_scatter_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ arg,
                                                               float* __restrict__ dx, bf16_t* __restrict__ dxb, int64_t groups, int s,
                                                               int d4) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (group, token, 4-channel column)
    if (e >= groups * s * d4) return;
    const int c = (int)(e % d4);
    const int t = (int)((e / d4) % s);
    const int64_t g = e / ((int64_t)d4 * s);
    const float4 v = reinterpret_cast<const float4*>(dout)[g * d4 + c];
    const uchar4 k = reinterpret_cast<const uchar4*>(arg)[g * d4 + c];
    const float4 o = make_float4(k.x == t ? v.x : 0.f, k.y == t ? v.y : 0.f, k.z == t ? v.z : 0.f, k.w == t ? v.w : 0.f);
    reinterpret_cast<float4*>(dx)[e] = o;
    if (dxb) store4(dxb + 4 * e, o);
}

}  // namespace pda

namespace pda {

template <typename TB>
static int launch_add_max_pool(const float* a, const TB* b, float* out, uint8_t* arg, int64_t groups, int seq, int d, hipStream_t st,
                               const char* what) {
    PDA_REQUIRE(groups >= 0 && seq >= 1 && seq <= 255 && d >= 4 && (d & 3) == 0, "%s: groups=%lld seq=%d d=%d", what, (long long)groups, seq, d);
    if (groups == 0) return PDA_OK;
    PDA_REQUIRE(a && b && out && arg, "%s: null pointer", what);
    PDA_REQUIRE((((uintptr_t)a | (uintptr_t)out) & 15) == 0 && ((uintptr_t)b & (4 * sizeof(TB) - 1)) == 0 && ((uintptr_t)arg & 3) == 0,
                "%s: alignment", what);
    const int64_t n = groups * (d / 4);
    hipLaunchKernelGGL(add_max_pool_kernel<TB>, dim3((unsigned)divup64(n, 256)), dim3(256), 0, st, a, b, out, arg, groups, seq, d / 4);
    return check_launch(what);
}

static int launch_max_pool_scatter(const float* grad_out, const uint8_t* arg, float* grad_x, bf16_t* grad_x_b, int64_t groups, int seq, int d,
                                   hipStream_t st, const char* what) {
    PDA_REQUIRE(groups >= 0 && seq >= 1 && seq <= 255 && d >= 4 && (d & 3) == 0, "%s: groups=%lld seq=%d d=%d", what, (long long)groups, seq, d);
    if (groups == 0) return PDA_OK;
    PDA_REQUIRE(grad_out && arg && grad_x, "%s: null pointer", what);
    PDA_REQUIRE((((uintptr_t)grad_out | (uintptr_t)grad_x) & 15) == 0 && ((uintptr_t)arg & 3) == 0 && ((uintptr_t)grad_x_b & 7) == 0,
                "%s: alignment", what);
    const int64_t n = groups * seq * (d / 4);
    hipLaunchKernelGGL(max_pool_scatter_kernel, dim3((unsigned)divup64(n, 256)), dim3(256), 0, st, grad_out, arg, grad_x, grad_x_b, groups,
                       seq, d / 4);
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_add_max_pool(const float* a, const float* b, float* out, uint8_t* arg, int64_t groups, int seq, int d,
                             pda_stream_t stream) {
    return pda::launch_add_max_pool<float>(a, b, out, arg, groups, seq, d, (hipStream_t)stream, "pda_add_max_pool");
}

PDA_API int pda_add_max_pool_bf16(const float* a, const uint16_t* b, float* out, uint8_t* arg, int64_t groups, int seq, int d,
                                  pda_stream_t stream) {
    return pda::launch_add_max_pool<pda::bf16_t>(a, b, out, arg, groups, seq, d, (hipStream_t)stream, "pda_add_max_pool_bf16");
}

PDA_API int pda_max_pool_scatter(const float* grad_out, const uint8_t* arg, float* grad_x, int64_t groups, int seq, int d,
                                 pda_stream_t stream) {
    return pda::launch_max_pool_scatter(grad_out, arg, grad_x, nullptr, groups, seq, d, (hipStream_t)stream, "pda_max_pool_scatter");
}

PDA_API int pda_max_pool_scatter_bf16(const float* grad_out, const uint8_t* arg, float* grad_x, uint16_t* grad_x_bf16, int64_t groups,
                                      int seq, int d, pda_stream_t stream) {
    PDA_REQUIRE(grad_x_bf16 != nullptr || groups == 0, "pda_max_pool_scatter_bf16: null pointer");
    return pda::launch_max_pool_scatter(grad_out, arg, grad_x, grad_x_bf16, groups, seq, d, (hipStream_t)stream, "pda_max_pool_scatter_bf16");
}
