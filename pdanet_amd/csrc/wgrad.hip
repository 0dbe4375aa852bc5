// wgrad.hip -- weight and bias gradient of a linear layer over a very long token axis
// (include/pda_train.h):  dW[n][m] = sum_t G[t][n] * X[t][m],  db[n] = sum_t G[t][n],
// X (T, M) activations, G (T, N) output gradients, both row-major, T = 65k..1M tokens, M, N <= 1536.
//
// In PDA-SSD's backward these are GEMMs with a tiny output and a huge reduction dimension
// (e.g. 256 x 128 output, K = 262144): the BLAS libraries reach 24-70 TFLOP/s on them even after tuning
// (tuning/tunableop_*.csv) because only a handful of output tiles exist, and the bias gradient is a
// separate reduction pass over G.  Here the reduction axis is split over the chip: a workgroup owns one
// 128 x 128 output tile and one slice of T, streams its slices of G and X through double-buffered LDS
// (whole 512-byte rows, 16-byte lanes), multiplies on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulate) with a 64 x 64 register tile per wave, and adds the column sums of G (the bias gradient)
// on the way.  Partials are written per slice and summed in fixed order by a second kernel
// (deterministic).  Workgroups that share a slice of T are placed on the same XCD so that the second
// reader of a G / X chunk hits that XCD's L2.
#include "pda_common.h"

namespace pda {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_KC = 32;       // token rows per LDS chunk
constexpr int WG_TILE = 128;    // output tile edge

__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                    float* __restrict__ part_w, float* __restrict__ part_b, int64_t T, int M,
                                                    int N, int tiles_m, int tiles_n, int S, int64_t KS, int nblocks) {
    __shared__ __attribute__((aligned(16))) float ldsG[2][WG_KC][WG_TILE];
    __shared__ __attribute__((aligned(16))) float ldsX[2][WG_KC][WG_TILE];
    // XCD-aware decode: hardware deals consecutive block ids round-robin over the 8 XCDs
    const int per_xcd = (nblocks + 7) / 8;
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= nblocks) return;
    const int tiles = tiles_m * tiles_n;
    const int tile = logical % tiles, s = logical / tiles;
    const int n0 = (tile / tiles_m) * WG_TILE, m0 = (tile % tiles_m) * WG_TILE;
    const int64_t t_begin = (int64_t)s * KS, t_end = (t_begin + KS < T) ? t_begin + KS : T;

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int c = lane & 31, h = lane >> 5;
    const int wn = w >> 1, wm = w & 1;
    // loader role: 4 passes of 8 rows; thread -> (row = tid / 32, 16-byte column = tid % 32)
    const int lrow = tid >> 5, lcol = (tid & 31) * 4;
    const bool g_ok = n0 + lcol < N, x_ok = m0 + lcol < M;   // N, M are multiples of 4: a float4 is in or out as a whole

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[2] = {0.f, 0.f};

    float4 rg[4], rx[4];
    // Loads are unconditional (a divergent branch around a load makes the compiler wait for it at the
    // branch's merge point, one load at a time): out-of-range rows / columns read a clamped address and
    // are zeroed by a select.
    const int gcol = g_ok ? n0 + lcol : 0, xcol = x_ok ? m0 + lcol : 0;
    auto load_chunk = [&](int64_t t0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int64_t t = t0 + p * 8 + lrow;
            const bool in = t < t_end;
            const int64_t tc = in ? t : t_end - 1;
            const float4 vg = *reinterpret_cast<const float4*>(G + tc * N + gcol);
            const float4 vx = *reinterpret_cast<const float4*>(X + tc * M + xcol);
            rg[p] = (in && g_ok) ? vg : make_float4(0.f, 0.f, 0.f, 0.f);
            rx[p] = (in && x_ok) ? vx : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<float4*>(&ldsG[buf][p * 8 + lrow][lcol]) = rg[p];
            *reinterpret_cast<float4*>(&ldsX[buf][p * 8 + lrow][lcol]) = rx[p];
        }
    };

    if (t_begin < t_end) {
        load_chunk(t_begin);
        store_chunk(0);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t t0 = t_begin; t0 < t_end; t0 += WG_KC) {
        const bool more = t0 + WG_KC < t_end;
        if (more) load_chunk(t0 + WG_KC);       // global loads in flight under the MFMAs
#pragma unroll
        for (int kk = 0; kk < WG_KC / 2; ++kk) {
            const int k = 2 * kk + h;
            const float a0 = ldsG[buf][k][wn * 64 + c], a1 = ldsG[buf][k][wn * 64 + 32 + c];
            const float b0 = ldsX[buf][k][wm * 64 + c], b1 = ldsX[buf][k][wm * 64 + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            bsum[0] += a0; bsum[1] += a1;
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // epilogue: D[row n][col m]: lane (c = m, h), register r -> n = (r & 3) + 8 * (r >> 2) + 4 * h
    float* pw = part_w + (size_t)s * N * M;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 64 + j * 32 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < N && m < M) pw[(size_t)n * M + m] = acc[i][j][r];
            }
        }
    if (part_b && m0 == 0 && wm == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float v = bsum[i] + __shfl_xor(bsum[i], 32);
            const int n = n0 + wn * 64 + i * 32 + c;
            if (h == 0 && n < N) part_b[(size_t)s * N + n] = v;
        }
    }
}

// second stage: out[e] = sum_s part[s][e], fixed order.  Block = 64 elements x 16 slices of S (independent loads
// in groups of 8: a thread walking all S partials alone would serialise S load latencies).
__device__ __forceinline__ float sum_slices(const float* __restrict__ part, int S, int64_t stride, int64_t e, bool live, int slice,
                                            float (*red)[64], int el) {
    float a = 0.f;
    if (live) {
        float v[8];
        for (int s0 = slice; s0 < S; s0 += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int sidx = s0 + 16 * u;
                v[u] = sidx < S ? part[(size_t)sidx * stride + e] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
    }
    red[slice][el] = a;
    __syncthreads();
    float r = 0.f;
    if (slice == 0)
#pragma unroll
        for (int p = 0; p < 16; ++p) r += red[p][el];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part_w, const float* __restrict__ part_b,
                                                            float* __restrict__ dw, float* __restrict__ db, int S, int64_t nm, int N) {
    __shared__ float red[16][64];
    const int el = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + el;
    const float w = sum_slices(part_w, S, nm, e, e < nm, slice, red, el);
    if (slice == 0 && e < nm) dw[e] = w;
    if (db && (int64_t)blockIdx.x * 64 < N) {          // block-uniform
        const float bsum = sum_slices(part_b, S, N, e, e < N, slice, red, el);
        if (slice == 0 && e < N) db[e] = bsum;
    }
}

static void wgrad_plan(int64_t T, int M, int N, int& tiles_m, int& tiles_n, int& S, int64_t& KS) {
    tiles_m = divup(M, WG_TILE); tiles_n = divup(N, WG_TILE);
    const int tiles = tiles_m * tiles_n;
    int64_t s = 512 / tiles;                       // ~2 workgroups per CU
    const int64_t max_s = T / 256 > 0 ? T / 256 : 1;  // at least 256 token rows per slice
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    KS = divup64(divup64(T, s), WG_KC) * WG_KC;
    S = (int)divup64(T, KS);
}

}  // namespace pda

PDA_API int64_t pda_linear_wgrad_scratch_bytes(int64_t tokens, int in_features, int out_features) {
    if (tokens <= 0 || in_features <= 0 || out_features <= 0) return 0;
    int tm, tn, S;
    int64_t KS;
    pda::wgrad_plan(tokens, in_features, out_features, tm, tn, S, KS);
    return (int64_t)S * ((int64_t)in_features * out_features + out_features) * (int64_t)sizeof(float);
}

PDA_API int pda_linear_wgrad(const float* x, const float* grad_out, float* grad_weight, float* grad_bias, void* scratch,
                             int64_t tokens, int in_features, int out_features, pda_stream_t stream) {
    const int M = in_features, N = out_features;
    PDA_REQUIRE(tokens >= 1 && M >= 4 && N >= 4 && (M & 3) == 0 && (N & 3) == 0,
                "pda_linear_wgrad: tokens=%lld in=%d out=%d (features must be multiples of 4)", (long long)tokens, M, N);
    PDA_REQUIRE(x && grad_out && grad_weight && scratch, "pda_linear_wgrad: null pointer");
    PDA_REQUIRE((((uintptr_t)x | (uintptr_t)grad_out) & 15) == 0, "pda_linear_wgrad: x / grad_out must be 16-byte aligned");
    int tm, tn, S;
    int64_t KS;
    pda::wgrad_plan(tokens, M, N, tm, tn, S, KS);
    const int nblocks = S * tm * tn;
    float* part_w = (float*)scratch;
    float* part_b = part_w + (size_t)S * N * M;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pda::wgrad_kernel, dim3(pda::divup(nblocks, 8) * 8), dim3(256), 0, st, x, grad_out, part_w,
                       grad_bias ? part_b : (float*)nullptr, tokens, M, N, tm, tn, S, KS, nblocks);
    const int64_t nm = (int64_t)N * M;
    hipLaunchKernelGGL(pda::wgrad_reduce_kernel, dim3((unsigned)pda::divup64(nm, 64)), dim3(1024), 0, st, part_w, part_b, grad_weight,
                       grad_bias, S, nm, N);
    return pda::check_launch("pda_linear_wgrad");
}
