// bn_relu.hip -- training-mode BatchNorm + ReLU over the LAST dimension of a (rows, C) tensor, forward and
// backward (include/pda_train.h).  The PDA / SA layers run point-major, so every
// Conv1x1 -> BatchNorm -> ReLU of the reference (pointnet2_modules.py:1605-1611, :628-671) is
// linear -> BN -> ReLU over rows = B*npoint*nsample.  Through torch that is MIOpen's spatial BN (3 kernels
// forward, 3 backward) plus separate ReLU forward/backward passes: 13 trips over the activation per layer.
// Here: forward = statistics pass + normalise/ReLU pass (2 reads, 1 write); backward = reduction pass +
// gradient pass (4 reads, 1 write); the ReLU mask is recomputed from x, never stored.
//
// Layout: C is a power of two, 4 <= C <= 1024; a thread owns 4 consecutive channels (16-byte accesses) and
// 256/(C/4) rows per step; per-channel sums are accumulated in double per thread, reduced over the block
// in LDS and written as per-block partials that a one-block kernel adds in fixed order (deterministic).
#include "pda_common.h"

namespace pda {

constexpr int BN_BLOCKS = 512;

struct BnShape {
    int64_t rows;
    int c, cg, rpb;  // cg = C/4 thread columns; rpb = rows per block step = 256 / cg (>= 1)
    int ns;          // pooled forms: rows per group (the max-pool runs over ns consecutive rows)
    // Multiplicity-weighted form (unique-token execution, csrc/ragged.hip): row r stands for roww[r] identical rows of
    // the dense tensor; the statistics are those of the dense tensor (count = its number of rows) and a row's gradient is
    // the SUM over its copies.  roww == nullptr: every weight is 1 and count == rows.
    const float* roww;
    int64_t count;
};

// Pooled backward: the incoming gradient is that of out[g][c] = max over the ns rows of group g of y; it reaches row
// g*ns + arg[g][c] only.  The dense (rows, C) gradient is never built: it is generated on the fly from (gout, arg).
__device__ __forceinline__ float4 pooled_grad(const float* __restrict__ gout, const uint8_t* __restrict__ arg, int64_t r,
                                              int col, const BnShape& s) {
    const int64_t g = r / s.ns;
    const int t = (int)(r - g * s.ns);
    const float4 v = reinterpret_cast<const float4*>(gout)[g * s.cg + col];
    const uchar4 k = reinterpret_cast<const uchar4*>(arg)[g * s.cg + col];
    return make_float4(k.x == t ? v.x : 0.f, k.y == t ? v.y : 0.f, k.z == t ? v.z : 0.f, k.w == t ? v.w : 0.f);
}

__device__ __forceinline__ void block_reduce_cols(double (&a)[4], double (&b)[4], double* lds, int cg, int rpb, double* out_a,
                                                  double* out_b) {
    // threads (r, col): reduce over r = tid / cg for every col = tid % cg; results for 4*cg channels
    const int tid = threadIdx.x;
    double* la = lds;             // [256][4]
    double* lb = lds + 256 * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { la[tid * 4 + k] = a[k]; lb[tid * 4 + k] = b[k]; }
    __syncthreads();
    if (tid < cg) {
        double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
        for (int r = 0; r < rpb; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) { sa[k] += la[(r * cg + tid) * 4 + k]; sb[k] += lb[(r * cg + tid) * 4 + k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { out_a[tid * 4 + k] = sa[k]; out_b[tid * 4 + k] = sb[k]; }
    }
}

// BWD == false: partial[blk][0][c] = sum x, [1][c] = sum x^2
// BWD == true : with dyh = dy * [(x - mean) * invstd * gamma + beta > 0]: [0][c] = sum dyh, [1][c] = sum dyh * xhat
// TX / TD: element types of x and dy -- float, or bf16 where the tensor is the output of a bf16 GEMM (dense-bf16 mode)
template <bool BWD, typename TX, typename TD, bool POOL = false>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const TX* __restrict__ x, const TD* __restrict__ dy,
                                                        const float* __restrict__ mean_invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, double* __restrict__ partial, BnShape s,
                                                        const uint8_t* __restrict__ arg = nullptr) {
    __shared__ double lds[2 * 256 * 4];
    const int tid = threadIdx.x;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (s.cg <= 256) {
        const int col = tid % s.cg, r0 = tid / s.cg;
        float4 mu = make_float4(0, 0, 0, 0), is = mu, g = mu, be = mu;
        if (BWD) {
            mu = reinterpret_cast<const float4*>(mean_invstd)[col];
            is = reinterpret_cast<const float4*>(mean_invstd + s.c)[col];
            g = reinterpret_cast<const float4*>(gamma)[col];
            be = reinterpret_cast<const float4*>(beta)[col];
        }
        // four strided rows per trip: four independent 16-byte loads (per operand) in flight per thread instead of one -- at one
        // load per thread the pass moves ~2.8 TB/s (Little: 512 workgroups x 256 threads x 16 B against ~1 us of latency)
        const int64_t stride = (int64_t)gridDim.x * s.rpb;
        auto row = [&](int64_t r, const float4& v, const float4& d) {
            if (!BWD) {
                const double w = s.roww ? (double)s.roww[r] : 1.0;
                a[0] += w * v.x; a[1] += w * v.y; a[2] += w * v.z; a[3] += w * v.w;
                b[0] += w * ((double)v.x * v.x); b[1] += w * ((double)v.y * v.y); b[2] += w * ((double)v.z * v.z); b[3] += w * ((double)v.w * v.w);
            } else {
                const float xh[4] = {(v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w};
                const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {be.x, be.y, be.z, be.w}, dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dyh = (xh[k] * gg[k] + bb[k] > 0.f) ? dd[k] : 0.f;
                    a[k] += dyh;
                    b[k] += (double)dyh * xh[k];
                }
            }
        };
        auto grad = [&](int64_t r) -> float4 {
            if constexpr (!BWD) return make_float4(0, 0, 0, 0);
            else if constexpr (POOL) return pooled_grad(reinterpret_cast<const float*>(dy), arg, r, col, s);
            else return load4(dy + r * s.c + 4 * col);
        };
        constexpr int U = BWD ? 1 : 4;      // the backward pass (two operands per row, more registers) measured no faster unrolled
        int64_t r = (int64_t)blockIdx.x * s.rpb + r0;
        for (; r + (U - 1) * stride < s.rows; r += U * stride) {
            float4 v[U], d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { v[u] = load4(x + (r + u * stride) * s.c + 4 * col); d[u] = grad(r + u * stride); }
#pragma unroll
            for (int u = 0; u < U; ++u) row(r + u * stride, v[u], d[u]);
        }
        for (; r < s.rows; r += stride) row(r, load4(x + r * s.c + 4 * col), grad(r));
    }
    double* pa = partial + (size_t)blockIdx.x * 2 * s.c;
    block_reduce_cols(a, b, lds, s.cg, s.rpb, pa, pa + s.c);
}

// finalize kernels: grid = ceil(C/32) blocks of 512 threads = 32 channels x 16 slices of the per-block partials
// (a single-thread-per-channel loop over 512 partials would serialise ~1000 dependent loads per layer)
__device__ __forceinline__ bool sum_partials(const double* __restrict__ partial, int nblocks, int c, double& s1, double& s2, int& ch) {
    __shared__ double red[2][16][32];
    const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
    ch = blockIdx.x * 32 + cl;
    double a = 0, b = 0;
    if (ch < c) {
        // BN_BLOCKS / 16 = 32 loads per sum and thread: two rounds of 16 independent loads 
        double va[16], vb[16];
        for (int k0 = part; k0 < nblocks; k0 += 16 * 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + 16 * u;
                va[u] = k < nblocks ? partial[(size_t)k * 2 * c + ch] : 0.0;
                vb[u] = k < nblocks ? partial[(size_t)k * 2 * c + c + ch] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { a += va[u]; b += vb[u]; }
        }
    }
    red[0][part][cl] = a; red[1][part][cl] = b;
    __syncthreads();
    if (part != 0 || ch >= c) return false;
    s1 = 0; s2 = 0;
#pragma unroll
    for (int p = 0; p < 16; ++p) { s1 += red[0][p][cl]; s2 += red[1][p][cl]; }
    return true;
}

// forward: mean / invstd (biased variance), running statistics (unbiased variance, momentum)
__global__ __launch_bounds__(512) void bn_finalize_fwd_kernel(const double* __restrict__ partial, int nblocks, int c, int64_t rows,
                                                               float eps, float momentum, float* __restrict__ mean_invstd,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var) {
    double s1, s2;
    int ch;
    if (!sum_partials(partial, nblocks, c, s1, s2, ch)) return;
    const double mean = s1 / (double)rows;
    double var = s2 / (double)rows - mean * mean;
    var = var < 0 ? 0 : var;
    mean_invstd[ch] = (float)mean;
    mean_invstd[c + ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = rows > 1 ? var * ((double)rows / (double)(rows - 1)) : var;
        running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
        running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
    }
}

// backward: dgamma = sum dyh*xhat, dbeta = sum dyh; sums[0][c] = dbeta / rows, sums[1][c] = dgamma / rows
__global__ __launch_bounds__(512) void bn_finalize_bwd_kernel(const double* __restrict__ partial, int nblocks, int c, int64_t rows,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ sums) {
    double s1, s2;
    int ch;
    if (!sum_partials(partial, nblocks, c, s1, s2, ch)) return;
    dbeta[ch] = (float)s1;
    dgamma[ch] = (float)s2;
    sums[ch] = (float)(s1 / (double)rows);
    sums[c + ch] = (float)(s2 / (double)rows);
}

// BWD == false: y = relu((x - mean) * invstd * gamma + beta)
// BWD == true : dx = gamma * invstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat))
template <bool BWD, typename TX, typename TD, typename TO, bool POOL = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TX* __restrict__ x, const TD* __restrict__ dy,
                                                       const float* __restrict__ mean_invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ sums,
                                                       TO* __restrict__ out, BnShape s, const uint8_t* __restrict__ arg = nullptr) {
    const int tid = threadIdx.x;
    const int col = tid % s.cg, r0 = tid / s.cg;
    const float4 mu = reinterpret_cast<const float4*>(mean_invstd)[col], is = reinterpret_cast<const float4*>(mean_invstd + s.c)[col];
    const float4 g = reinterpret_cast<const float4*>(gamma)[col], be = reinterpret_cast<const float4*>(beta)[col];
    float4 m1 = make_float4(0, 0, 0, 0), m2 = m1;
    if (BWD) { m1 = reinterpret_cast<const float4*>(sums)[col]; m2 = reinterpret_cast<const float4*>(sums + s.c)[col]; }
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w}, gv[4] = {g.x, g.y, g.z, g.w},
                bv[4] = {be.x, be.y, be.z, be.w}, m1v[4] = {m1.x, m1.y, m1.z, m1.w}, m2v[4] = {m2.x, m2.y, m2.z, m2.w};
    const int64_t stride = (int64_t)gridDim.x * s.rpb;
    auto row = [&](int64_t r, const float4& v, const float4& d) {
        const float xv[4] = {v.x, v.y, v.z, v.w};
        float o[4];
        if (!BWD) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaxf((xv[k] - muv[k]) * isv[k] * gv[k] + bv[k], 0.f);
        } else {
            const float dv[4] = {d.x, d.y, d.z, d.w};
            const float w = s.roww ? s.roww[r] : 1.f;      // the copies of a row share the mean terms, the incoming gradient is their sum
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xv[k] - muv[k]) * isv[k];
                const float dyh = (xh * gv[k] + bv[k] > 0.f) ? dv[k] : 0.f;
                o[k] = gv[k] * isv[k] * (dyh - w * (m1v[k] + xh * m2v[k]));
            }
        }
        store4(out + r * s.c + 4 * col, make_float4(o[0], o[1], o[2], o[3]));
    };
    auto grad = [&](int64_t r) -> float4 {
        if constexpr (!BWD) return make_float4(0, 0, 0, 0);
        else if constexpr (POOL) return pooled_grad(reinterpret_cast<const float*>(dy), arg, r, col, s);
        else return load4(dy + r * s.c + 4 * col);
    };
    constexpr int U = (BWD && !POOL) ? 1 : 4;           // several strided rows per trip: see bn_reduce_kernel
    // The rows are walked from the LAST to the first: the statistics pass before this one swept them first to last, so
    // what the 256 MB Infinity Cache still holds is the tail of the tensors -- in the same order a tensor larger than
    // the cache would be re-read from HBM in full.
    const int64_t last = s.rows - 1;
    int64_t r = (int64_t)blockIdx.x * s.rpb + r0;
    for (; r + (U - 1) * stride < s.rows; r += U * stride) {
        float4 v[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t q = last - (r + u * stride); v[u] = load4(x + q * s.c + 4 * col); d[u] = grad(q); }
#pragma unroll
        for (int u = 0; u < U; ++u) row(last - (r + u * stride), v[u], d[u]);
    }
    for (; r < s.rows; r += stride) { const int64_t q = last - r; row(q, load4(x + q * s.c + 4 * col), grad(q)); }
}

// forward with the max-pool fused in: thread = (group, 4-channel column); y = relu(bn(x)) is never written, out (groups, C)
// = max over the group's ns rows, arg = the first row that attains it
template <typename TX>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const TX* __restrict__ x, const float* __restrict__ mean_invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ out, uint8_t* __restrict__ arg, int64_t groups,
                                                            BnShape s) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= groups * s.cg) return;
    const int64_t g = e / s.cg;
    const int col = (int)(e - g * s.cg);
    const float4 mu = reinterpret_cast<const float4*>(mean_invstd)[col], is = reinterpret_cast<const float4*>(mean_invstd + s.c)[col];
    const float4 gm = reinterpret_cast<const float4*>(gamma)[col], be = reinterpret_cast<const float4*>(beta)[col];
    float4 best = make_float4(-1.f, -1.f, -1.f, -1.f);     // y >= 0: the first row always wins against -1
    uchar4 bi = make_uchar4(0, 0, 0, 0);
    const TX* px = x + (size_t)g * s.ns * s.c + 4 * col;
    for (int t = 0; t < s.ns; ++t) {
        const float4 v = load4(px + (size_t)t * s.c);
        const float y0 = fmaxf((v.x - mu.x) * is.x * gm.x + be.x, 0.f), y1 = fmaxf((v.y - mu.y) * is.y * gm.y + be.y, 0.f);
        const float y2 = fmaxf((v.z - mu.z) * is.z * gm.z + be.z, 0.f), y3 = fmaxf((v.w - mu.w) * is.w * gm.w + be.w, 0.f);
        if (y0 > best.x) { best.x = y0; bi.x = (uint8_t)t; }
        if (y1 > best.y) { best.y = y1; bi.y = (uint8_t)t; }
        if (y2 > best.z) { best.z = y2; bi.z = (uint8_t)t; }
        if (y3 > best.w) { best.w = y3; bi.w = (uint8_t)t; }
    }
    reinterpret_cast<float4*>(out)[e] = best;
    reinterpret_cast<uchar4*>(arg)[e] = bi;
}

static int bn_shape(int64_t rows, int c, BnShape& s, const char* what) {
    PDA_REQUIRE(rows >= 1, "%s: rows = %lld", what, (long long)rows);
    PDA_REQUIRE(c >= 4 && c <= 1024 && (c & (c - 1)) == 0, "%s: C = %d is not a power of two in [4, 1024]", what, c);
    s.rows = rows; s.c = c; s.cg = c / 4; s.rpb = 256 / s.cg; s.ns = 1;
    s.roww = nullptr; s.count = rows;
    return PDA_OK;
}

static int bn_grid(const BnShape& s) {
    const int64_t steps = divup64(s.rows, s.rpb);
    return (int)(steps < BN_BLOCKS ? steps : BN_BLOCKS);
}

}  // namespace pda

PDA_API int64_t pda_bn_relu_scratch_bytes(int c) {
    const int64_t cc = c > 0 ? c : 0;
    return (int64_t)pda::BN_BLOCKS * 2 * cc * (int64_t)sizeof(double) + 2 * cc * (int64_t)sizeof(float);
}

namespace pda {

template <typename T> static bool bn_aligned(const T* p) { return ((uintptr_t)p & (4 * sizeof(T) - 1)) == 0; }

template <typename TX, typename TY>
static int launch_bn_relu_fwd(const TX* x, const float* gamma, const float* beta, float* running_mean, float* running_var, TY* y,
                              float* mean_invstd, void* scratch, int64_t rows, int c, float eps, float momentum, hipStream_t st,
                              const char* what, const float* roww = nullptr, int64_t count = 0) {
    BnShape s;
    if (int rc = bn_shape(rows, c, s, what)) return rc;
    if (roww) { PDA_REQUIRE(count >= rows, "%s: count = %lld < rows", what, (long long)count); s.roww = roww; s.count = count; }
    PDA_REQUIRE(x && gamma && beta && y && mean_invstd && scratch, "%s: null pointer", what);
    PDA_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "%s: running_mean/var must come together", what);
    PDA_REQUIRE(bn_aligned(x) && bn_aligned(y) && (((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mean_invstd) & 15) == 0,
                "%s: pointers must be 16-byte aligned (8 for bf16 tensors)", what);
    const int grid = bn_grid(s);
    hipLaunchKernelGGL((bn_reduce_kernel<false, TX, float>), dim3(grid), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (double*)scratch, s);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(divup(c, 32)), dim3(512), 0, st, (const double*)scratch, grid, c, s.count, eps, momentum,
                       mean_invstd, running_mean, running_var);
    hipLaunchKernelGGL((bn_apply_kernel<false, TX, float, TY>), dim3(grid), dim3(256), 0, st, x, (const float*)nullptr, mean_invstd, gamma, beta,
                       (const float*)nullptr, y, s);
    return check_launch(what);
}

template <typename TX, typename TD>
static int launch_bn_relu_bwd(const TX* x, const TD* grad_y, const float* gamma, const float* beta, const float* mean_invstd, TX* grad_x,
                              float* grad_gamma, float* grad_beta, void* scratch, int64_t rows, int c, hipStream_t st, const char* what,
                              const float* roww = nullptr, int64_t count = 0) {
    BnShape s;
    if (int rc = bn_shape(rows, c, s, what)) return rc;
    if (roww) { PDA_REQUIRE(count >= rows, "%s: count = %lld < rows", what, (long long)count); s.roww = roww; s.count = count; }
    PDA_REQUIRE(x && grad_y && gamma && beta && mean_invstd && grad_x && grad_gamma && grad_beta && scratch, "%s: null pointer", what);
    PDA_REQUIRE(bn_aligned(x) && bn_aligned(grad_y) && bn_aligned(grad_x) && (((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mean_invstd) & 15) == 0,
                "%s: pointers must be 16-byte aligned (8 for bf16 tensors)", what);
    const int grid = bn_grid(s);
    // the per-channel means of the second pass live behind the partials in the scratch buffer
    float* sums = reinterpret_cast<float*>(reinterpret_cast<double*>(scratch) + (size_t)BN_BLOCKS * 2 * c);
    hipLaunchKernelGGL((bn_reduce_kernel<true, TX, TD>), dim3(grid), dim3(256), 0, st, x, grad_y, mean_invstd, gamma, beta, (double*)scratch, s);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(divup(c, 32)), dim3(512), 0, st, (const double*)scratch, grid, c, s.count, grad_gamma, grad_beta,
                       sums);
    hipLaunchKernelGGL((bn_apply_kernel<true, TX, TD, TX>), dim3(grid), dim3(256), 0, st, x, grad_y, mean_invstd, gamma, beta, sums, grad_x, s);
    return check_launch(what);
}

template <typename TX>
static int launch_bn_relu_pool_fwd(const TX* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* out,
                                   uint8_t* arg, float* mean_invstd, void* scratch, int64_t groups, int ns, int c, float eps,
                                   float momentum, hipStream_t st, const char* what) {
    PDA_REQUIRE(groups >= 1 && ns >= 1 && ns <= 255, "%s: groups=%lld ns=%d (1..255)", what, (long long)groups, ns);
    BnShape s;
    if (int rc = bn_shape(groups * ns, c, s, what)) return rc;
    s.ns = ns;
    PDA_REQUIRE(x && gamma && beta && out && arg && mean_invstd && scratch, "%s: null pointer", what);
    PDA_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "%s: running_mean/var must come together", what);
    PDA_REQUIRE(bn_aligned(x) && (((uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mean_invstd) & 15) == 0 &&
                    ((uintptr_t)arg & 3) == 0, "%s: alignment", what);
    const int grid = bn_grid(s);
    hipLaunchKernelGGL((bn_reduce_kernel<false, TX, float>), dim3(grid), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (double*)scratch, s, (const uint8_t*)nullptr);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(divup(c, 32)), dim3(512), 0, st, (const double*)scratch, grid, c, s.rows, eps, momentum,
                       mean_invstd, running_mean, running_var);
    hipLaunchKernelGGL(bn_apply_pool_kernel<TX>, dim3((unsigned)divup64(groups * s.cg, 256)), dim3(256), 0, st, x, mean_invstd, gamma, beta, out,
                       arg, groups, s);
    return check_launch(what);
}

template <typename TX>
static int launch_bn_relu_pool_bwd(const TX* x, const float* grad_out, const uint8_t* arg, const float* gamma, const float* beta,
                                   const float* mean_invstd, TX* grad_x, float* grad_gamma, float* grad_beta, void* scratch, int64_t groups,
                                   int ns, int c, hipStream_t st, const char* what) {
    PDA_REQUIRE(groups >= 1 && ns >= 1 && ns <= 255, "%s: groups=%lld ns=%d (1..255)", what, (long long)groups, ns);
    BnShape s;
    if (int rc = bn_shape(groups * ns, c, s, what)) return rc;
    s.ns = ns;
    PDA_REQUIRE(x && grad_out && arg && gamma && beta && mean_invstd && grad_x && grad_gamma && grad_beta && scratch, "%s: null pointer", what);
    PDA_REQUIRE(bn_aligned(x) && bn_aligned(grad_x) && (((uintptr_t)grad_out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mean_invstd) & 15) == 0 &&
                    ((uintptr_t)arg & 3) == 0, "%s: alignment", what);
    const int grid = bn_grid(s);
    float* sums = reinterpret_cast<float*>(reinterpret_cast<double*>(scratch) + (size_t)BN_BLOCKS * 2 * c);
    hipLaunchKernelGGL((bn_reduce_kernel<true, TX, float, true>), dim3(grid), dim3(256), 0, st, x, grad_out, mean_invstd, gamma, beta,
                       (double*)scratch, s, arg);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(divup(c, 32)), dim3(512), 0, st, (const double*)scratch, grid, c, s.rows, grad_gamma, grad_beta,
                       sums);
    hipLaunchKernelGGL((bn_apply_kernel<true, TX, float, TX, true>), dim3(grid), dim3(256), 0, st, x, grad_out, mean_invstd, gamma, beta, sums,
                       grad_x, s, arg);
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_bn_relu_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float* y, float* mean_invstd, void* scratch, int64_t rows, int c, float eps, float momentum,
                            pda_stream_t stream) {
    return pda::launch_bn_relu_fwd<float, float>(x, gamma, beta, running_mean, running_var, y, mean_invstd, scratch, rows, c, eps, momentum,
                                                 (hipStream_t)stream, "pda_bn_relu_fwd");
}

PDA_API int pda_bn_relu_bwd(const float* x, const float* grad_y, const float* gamma, const float* beta, const float* mean_invstd,
                            float* grad_x, float* grad_gamma, float* grad_beta, void* scratch, int64_t rows, int c,
                            pda_stream_t stream) {
    return pda::launch_bn_relu_bwd<float, float>(x, grad_y, gamma, beta, mean_invstd, grad_x, grad_gamma, grad_beta, scratch, rows, c,
                                                 (hipStream_t)stream, "pda_bn_relu_bwd");
}

PDA_API int pda_bn_relu_fwd_weighted(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     float* y, float* mean_invstd, void* scratch, int64_t rows, int c, float eps, float momentum,
                                     const float* row_weight, int64_t count, pda_stream_t stream) {
    PDA_REQUIRE(row_weight != nullptr, "pda_bn_relu_fwd_weighted: null pointer");
    return pda::launch_bn_relu_fwd<float, float>(x, gamma, beta, running_mean, running_var, y, mean_invstd, scratch, rows, c, eps, momentum,
                                                 (hipStream_t)stream, "pda_bn_relu_fwd_weighted", row_weight, count);
}

PDA_API int pda_bn_relu_bwd_weighted(const float* x, const float* grad_y, const float* gamma, const float* beta, const float* mean_invstd,
                                     float* grad_x, float* grad_gamma, float* grad_beta, void* scratch, int64_t rows, int c,
                                     const float* row_weight, int64_t count, pda_stream_t stream) {
    PDA_REQUIRE(row_weight != nullptr, "pda_bn_relu_bwd_weighted: null pointer");
    return pda::launch_bn_relu_bwd<float, float>(x, grad_y, gamma, beta, mean_invstd, grad_x, grad_gamma, grad_beta, scratch, rows, c,
                                                 (hipStream_t)stream, "pda_bn_relu_bwd_weighted", row_weight, count);
}

PDA_API int pda_bn_relu_fwd_mixed(const void* x, int x_is_bf16, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, void* y, int y_is_bf16, float* mean_invstd, void* scratch, int64_t rows, int c,
                                  float eps, float momentum, pda_stream_t stream) {
    using pda::bf16_t;
    hipStream_t st = (hipStream_t)stream;
    const char* what = "pda_bn_relu_fwd_mixed";
#define PDA_BN_FWD(TX, TY) pda::launch_bn_relu_fwd<TX, TY>((const TX*)x, gamma, beta, running_mean, running_var, (TY*)y, mean_invstd, scratch, rows, c, eps, momentum, st, what)
    if (x_is_bf16) return y_is_bf16 ? PDA_BN_FWD(bf16_t, bf16_t) : PDA_BN_FWD(bf16_t, float);
    return y_is_bf16 ? PDA_BN_FWD(float, bf16_t) : PDA_BN_FWD(float, float);
#undef PDA_BN_FWD
}

PDA_API int pda_bn_relu_bwd_mixed(const void* x, int x_is_bf16, const void* grad_y, int grad_y_is_bf16, const float* gamma,
                                  const float* beta, const float* mean_invstd, void* grad_x, float* grad_gamma, float* grad_beta,
                                  void* scratch, int64_t rows, int c, pda_stream_t stream) {
    using pda::bf16_t;
    hipStream_t st = (hipStream_t)stream;
    const char* what = "pda_bn_relu_bwd_mixed";
#define PDA_BN_BWD(TX, TD) pda::launch_bn_relu_bwd<TX, TD>((const TX*)x, (const TD*)grad_y, gamma, beta, mean_invstd, (TX*)grad_x, grad_gamma, grad_beta, scratch, rows, c, st, what)
    if (x_is_bf16) return grad_y_is_bf16 ? PDA_BN_BWD(bf16_t, bf16_t) : PDA_BN_BWD(bf16_t, float);
    return grad_y_is_bf16 ? PDA_BN_BWD(float, bf16_t) : PDA_BN_BWD(float, float);
#undef PDA_BN_BWD
}

PDA_API int pda_bn_relu_max_pool_fwd(const void* x, int x_is_bf16, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float* out, uint8_t* arg, float* mean_invstd, void* scratch, int64_t groups,
                                     int ns, int c, float eps, float momentum, pda_stream_t stream) {
    const char* what = "pda_bn_relu_max_pool_fwd";
    if (x_is_bf16)
        return pda::launch_bn_relu_pool_fwd<pda::bf16_t>((const pda::bf16_t*)x, gamma, beta, running_mean, running_var, out, arg, mean_invstd,
                                                         scratch, groups, ns, c, eps, momentum, (hipStream_t)stream, what);
    return pda::launch_bn_relu_pool_fwd<float>((const float*)x, gamma, beta, running_mean, running_var, out, arg, mean_invstd, scratch, groups,
                                               ns, c, eps, momentum, (hipStream_t)stream, what);
}

PDA_API int pda_bn_relu_max_pool_bwd(const void* x, int x_is_bf16, const float* grad_out, const uint8_t* arg, const float* gamma,
                                     const float* beta, const float* mean_invstd, void* grad_x, float* grad_gamma, float* grad_beta,
                                     void* scratch, int64_t groups, int ns, int c, pda_stream_t stream) {
    const char* what = "pda_bn_relu_max_pool_bwd";
    if (x_is_bf16)
        return pda::launch_bn_relu_pool_bwd<pda::bf16_t>((const pda::bf16_t*)x, grad_out, arg, gamma, beta, mean_invstd, (pda::bf16_t*)grad_x,
                                                         grad_gamma, grad_beta, scratch, groups, ns, c, (hipStream_t)stream, what);
    return pda::launch_bn_relu_pool_bwd<float>((const float*)x, grad_out, arg, gamma, beta, mean_invstd, (float*)grad_x, grad_gamma, grad_beta,
                                               scratch, groups, ns, c, (hipStream_t)stream, what);
}

// ---- the passes one at a time: chains whose statistics come out of a GEMM's epilogue (pda_gemm_split_bn, csrc/gemm_split.hip) and
// whose normalised activations are only ever formed in a consumer's operand load --------------------------------------------
PDA_API int pda_bn_stats_fwd(const float* x, float* running_mean, float* running_var, float* mean_invstd, void* scratch, int64_t rows, int c,
                             float eps, float momentum, pda_stream_t stream) {
    const char* what = "pda_bn_stats_fwd";
    pda::BnShape s;
    if (int rc = pda::bn_shape(rows, c, s, what)) return rc;
    PDA_REQUIRE(x && mean_invstd && scratch && pda::bn_aligned(x) && ((uintptr_t)mean_invstd & 15) == 0, "%s: null or misaligned pointer", what);
    PDA_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "%s: running_mean/var must come together", what);
    const int grid = pda::bn_grid(s);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((pda::bn_reduce_kernel<false, float, float>), dim3(grid), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (double*)scratch, s);
    hipLaunchKernelGGL(pda::bn_finalize_fwd_kernel, dim3(pda::divup(c, 32)), dim3(512), 0, st, (const double*)scratch, grid, c, s.count, eps, momentum,
                       mean_invstd, running_mean, running_var);
    return pda::check_launch(what);
}

PDA_API int pda_bn_finalize_fwd(const double* partial, int nblocks, int c, int64_t count, float eps, float momentum, float* mean_invstd,
                                float* running_mean, float* running_var, pda_stream_t stream) {
    PDA_REQUIRE(partial && mean_invstd && nblocks >= 1 && c >= 1 && count >= 1, "pda_bn_finalize_fwd: bad argument");
    PDA_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "pda_bn_finalize_fwd: running_mean/var must come together");
    hipLaunchKernelGGL(pda::bn_finalize_fwd_kernel, dim3(pda::divup(c, 32)), dim3(512), 0, (hipStream_t)stream, partial, nblocks, c, count, eps,
                       momentum, mean_invstd, running_mean, running_var);
    return pda::check_launch("pda_bn_finalize_fwd");
}

// the second pass of pda_bn_relu_max_pool_fwd alone (statistics given)
PDA_API int pda_bn_relu_max_pool_apply(const float* x, const float* gamma, const float* beta, const float* mean_invstd, float* out, uint8_t* arg,
                                       int64_t groups, int ns, int c, pda_stream_t stream) {
    const char* what = "pda_bn_relu_max_pool_apply";
    PDA_REQUIRE(groups >= 1 && ns >= 1 && ns <= 255, "%s: groups=%lld ns=%d (1..255)", what, (long long)groups, ns);
    pda::BnShape s;
    if (int rc = pda::bn_shape(groups * ns, c, s, what)) return rc;
    s.ns = ns;
    PDA_REQUIRE(x && gamma && beta && out && arg && mean_invstd, "%s: null pointer", what);
    PDA_REQUIRE(pda::bn_aligned(x) && (((uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mean_invstd) & 15) == 0 &&
                    ((uintptr_t)arg & 3) == 0, "%s: alignment", what);
    hipLaunchKernelGGL(pda::bn_apply_pool_kernel<float>, dim3((unsigned)pda::divup64(groups * s.cg, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       mean_invstd, gamma, beta, out, arg, groups, s);
    return pda::check_launch(what);
}
