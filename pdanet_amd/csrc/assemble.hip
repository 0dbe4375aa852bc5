// assemble.hip -- token assembly of a PDA scale (include/pda_train.h).
//
// pointnet2_modules.py:879-922 builds the encoder input per (centre, neighbour) token as
//   [position encoding | grouped feature * density score | grouped feature | centre's global feature]   (4C channels)
// through a grouping op (materialising the grouped features), a multiply, an expand and a concatenation, and
// autograd walks the same chain back (slice copies, two multiplies, an add, two reductions, a scatter-add).
// Point-major, both directions are one kernel each: the neighbour's feature row is gathered straight from the
// (B, N, C) table (it stays in L2), the (tokens, 4C) tensor is written / read exactly once.
#include "pda_common.h"

namespace pda {

// thread = (token, 4-channel column); C4 = C / 4
__global__ __launch_bounds__(256) void assemble_fwd_kernel(const float* __restrict__ rppe, const float* __restrict__ dscale,
                                                           const float* __restrict__ feats, const int* __restrict__ idx,
                                                           const float* __restrict__ glob, float* __restrict__ out, int n, int m,
                                                           int ns, int c4, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % c4);
    const int64_t tok = e / c4;
    const int64_t bm = tok / ns;
    const int64_t b = bm / m;
    const float4 r = reinterpret_cast<const float4*>(rppe)[e];
    const float4 f = reinterpret_cast<const float4*>(feats)[((size_t)b * n + idx[tok]) * c4 + c];
    const float4 g = reinterpret_cast<const float4*>(glob)[bm * c4 + c];
    const float d = dscale[tok];
    float4* o = reinterpret_cast<float4*>(out) + (size_t)tok * 4 * c4 + c;
    o[0] = r;
    o[c4] = make_float4(f.x * d, f.y * d, f.z * d, f.w * d);
    o[2 * c4] = f;
    o[3 * c4] = g;
}

// thread = (centre, 4-channel column), the C4 columns of a centre on consecutive lanes (C4 in {4, 8, 16, 32, 64}); the
// thread walks the centre's ns tokens.  The feature gradient is a scatter-add over neighbour rows (float atomics, as the
// reference's group_points_grad); ball query pads a short neighbour list by repeating its FIRST entry
// (ball_query_gpu.cu:40-46), and on LiDAR-like scenes 35-65 % of the tokens are such repeats: their contributions are
// summed in registers and leave as ONE atomic per channel, which also removes the worst same-address contention
// (measured: the atomics were 0.85 ms of the kernel's 1.12 ms per step).  d(glob) is a plain register sum.
template <int c4>
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ dscale,
                                                           const float* __restrict__ feats, const int* __restrict__ idx,
                                                           float* __restrict__ d_rppe, float* __restrict__ d_dscale,
                                                           float* __restrict__ d_feats, float* __restrict__ d_glob, int n, int m,
                                                           int ns, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (centre, column)
    const bool live = e < total;
    const int64_t ec = live ? e : total - 1;                     // dead lanes shadow the last element (shuffles stay uniform)
    const int c = (int)(ec % c4);
    const int64_t bm = ec / c4;
    const int64_t b = bm / m;
    const int64_t tok0 = bm * ns;
    const int first = idx[tok0];
    const float4* frows = reinterpret_cast<const float4*>(feats) + (size_t)b * n * c4 + c;
    float4 acc_g = make_float4(0.f, 0.f, 0.f, 0.f);      // d(glob)
    float4 acc_0 = make_float4(0.f, 0.f, 0.f, 0.f);      // feature gradient of the first neighbour and its repeats
    for (int t = 0; t < ns; ++t) {
        const int64_t tok = tok0 + t;
        const float4* g4 = reinterpret_cast<const float4*>(dx) + (size_t)tok * 4 * c4 + c;
        const float4 g_r = g4[0], g_fd = g4[c4], g_f = g4[2 * c4], g_g = g4[3 * c4];
        const int row = idx[tok];
        const float4 f = frows[(size_t)row * c4];
        const float d = dscale[tok];
        // d(dscale) = sum over the token's channels of g_fd * f: butterfly over the c4 lanes of the centre
        float part = (g_fd.x * f.x + g_fd.y * f.y) + (g_fd.z * f.z + g_fd.w * f.w);
#pragma unroll
        for (int o = c4 >> 1; o >= 1; o >>= 1) part += __shfl_xor(part, o);
        acc_g.x += g_g.x; acc_g.y += g_g.y; acc_g.z += g_g.z; acc_g.w += g_g.w;
        const float4 v = make_float4(g_fd.x * d + g_f.x, g_fd.y * d + g_f.y, g_fd.z * d + g_f.z, g_fd.w * d + g_f.w);
        const bool rep = row == first;
        acc_0.x += rep ? v.x : 0.f; acc_0.y += rep ? v.y : 0.f; acc_0.z += rep ? v.z : 0.f; acc_0.w += rep ? v.w : 0.f;
        if (live) {
            reinterpret_cast<float4*>(d_rppe)[(size_t)tok * c4 + c] = g_r;
            if (c == 0) d_dscale[tok] = part;
            if (!rep) {
                float* df = d_feats + (((size_t)b * n + row) * c4 + c) * 4;
                atomicAdd(df + 0, v.x); atomicAdd(df + 1, v.y); atomicAdd(df + 2, v.z); atomicAdd(df + 3, v.w);
            }
        }
    }
    if (!live) return;
    float* df = d_feats + (((size_t)b * n + first) * c4 + c) * 4;
    atomicAdd(df + 0, acc_0.x); atomicAdd(df + 1, acc_0.y); atomicAdd(df + 2, acc_0.z); atomicAdd(df + 3, acc_0.w);
    reinterpret_cast<float4*>(d_glob)[bm * c4 + c] = acc_g;
}

static bool assemble_c_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128 || c == 256; }

}  // namespace pda

PDA_API int pda_assemble_tokens(const float* rppe, const float* dscale, const float* feats, const int32_t* idx, const float* glob,
                                float* out, int b, int n, int m, int nsample, int c, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 1 && pda::assemble_c_ok(c), "pda_assemble_tokens: b=%d n=%d m=%d nsample=%d C=%d",
                b, n, m, nsample, c);
    const int64_t total = (int64_t)b * m * nsample * (c / 4);
    if (total == 0) return PDA_OK;
    PDA_REQUIRE(rppe && dscale && feats && idx && glob && out, "pda_assemble_tokens: null pointer");
    PDA_REQUIRE((((uintptr_t)rppe | (uintptr_t)feats | (uintptr_t)glob | (uintptr_t)out) & 15) == 0, "pda_assemble_tokens: alignment");
    hipLaunchKernelGGL(pda::assemble_fwd_kernel, dim3((unsigned)pda::divup64(total, 256)), dim3(256), 0, (hipStream_t)stream, rppe, dscale,
                       feats, idx, glob, out, n, m, nsample, c / 4, total);
    return pda::check_launch("pda_assemble_tokens");
}

PDA_API int pda_assemble_tokens_grad(const float* grad_out, const float* dscale, const float* feats, const int32_t* idx,
                                     float* grad_rppe, float* grad_dscale, float* grad_feats, float* grad_glob, int b, int n, int m,
                                     int nsample, int c, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 1 && pda::assemble_c_ok(c), "pda_assemble_tokens_grad: b=%d n=%d m=%d nsample=%d C=%d",
                b, n, m, nsample, c);
    const int64_t total = (int64_t)b * m * nsample * (c / 4);
    if (total == 0) return PDA_OK;
    PDA_REQUIRE(grad_out && dscale && feats && idx && grad_rppe && grad_dscale && grad_feats && grad_glob, "pda_assemble_tokens_grad: null pointer");
    PDA_REQUIRE((((uintptr_t)grad_out | (uintptr_t)feats | (uintptr_t)grad_rppe | (uintptr_t)grad_glob) & 15) == 0, "pda_assemble_tokens_grad: alignment");
    const int64_t centre_cols = (int64_t)b * m * (c / 4);
    const dim3 grid((unsigned)pda::divup64(centre_cols, 256)), block(256);
#define PDA_ASM_BWD(C4)                                                                                                               \
    case C4: hipLaunchKernelGGL(pda::assemble_bwd_kernel<C4>, grid, block, 0, (hipStream_t)stream, grad_out, dscale, feats, idx, grad_rppe, \
                                grad_dscale, grad_feats, grad_glob, n, m, nsample, centre_cols); break
    switch (c / 4) { PDA_ASM_BWD(4); PDA_ASM_BWD(8); PDA_ASM_BWD(16); PDA_ASM_BWD(32); PDA_ASM_BWD(64); }
#undef PDA_ASM_BWD
    return pda::check_launch("pda_assemble_tokens_grad");
}
