// ragged.hip -- unique-token ("ragged") execution of a PDA scale (include/pda_train.h).
//
// ball_query pads a neighbour list that found fewer than nsample points by repeating its FIRST entry
// (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:35-41; a list with no hit at all stays
// the caller's zeros = point 0 repeated, pointnet2_utils.py:246).  The PDA layer then runs its transformer encoder
// (pointnet2_modules.py:924-931, PointFormer.py:28-38) over ALL nsample tokens of every centre -- but a repeated
// neighbour is an identical token: same LayerNorm, same projections, same FFN row, same attention output.  With
// centres chosen by FPS the neighbourhoods are small: on the ONCE-shaped scenes of SURVEY 8(d) only 6-37 % of the
// (centre, neighbour) tokens are distinct.
//
// Here a scale is executed on its DISTINCT tokens only, bit-compatible in exact arithmetic with the dense form:
//   * every per-token operator (LayerNorm, the projections, the FFN) runs on the compact (U, D) matrix, U = number
//     of distinct tokens; group g owns rows [off[g], off[g] + cnt[g]);
//   * the softmax over the keys of a group sees key 0 with weight (nsample - cnt + 1): csrc/group_attention.hip adds
//     log(multiplicity) to that score, which is exactly the sum over the repeated keys;
//   * the max over the tokens of a group does not care about repeats;
//   * gradients: all copies of a token carry the same activations, so the sum of their gradients is what the compact
//     token receives; nothing else changes (the weighted key already yields summed dK / dV).
// This file: the plan (cnt / off / compact-row -> dense-row map), the token assembly and the add + max-pool tail in
// compact form.  No host synchronisation in here; the caller reads U (off[groups]) once per layer to size the GEMMs.
#include "pda_common.h"

namespace pda {

// cnt[g] = the distinct entries of group g (1 + number of entries that differ from the first: real hits are distinct and
// ascending, padding repeats entry 0).  One thread per group over the whole grid (a single workgroup walking 32 768 lists of
// config 5 took 250 us in front of the layer's host read).
__global__ __launch_bounds__(256) void ragged_count_kernel(const int32_t* __restrict__ idx, int32_t* __restrict__ cnt, int groups, int ns) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const int32_t* row = idx + (size_t)g * ns;
    const int32_t first = row[0];
    int32_t c = 1;
    for (int s = 1; s < ns; ++s) c += row[s] != first ? 1 : 0;
    cnt[g] = c;
}

// One workgroup: exclusive scan of cnt over all groups.  Thread t owns the `per` consecutive groups starting at t * per.
// off has groups + 1 entries; off[groups] = U.
__global__ __launch_bounds__(1024) void ragged_plan_kernel(const int32_t* __restrict__ cnt, int32_t* __restrict__ off, int groups) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int per = (groups + 1023) / 1024;
    const int g0 = t * per, g1 = min(groups, g0 + per);
    int32_t sum = 0;
    for (int g = g0; g < g1; ++g) sum += cnt[g];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {        // Hillis-Steele inclusive scan of the 1024 partial sums
        const int32_t v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int32_t run = part[t] - sum;                 // exclusive prefix of my first group
    for (int g = g0; g < g1; ++g) { off[g] = run; run += cnt[g]; }
    if (t == 1023) off[groups] = part[1023];
}

// rowmap[off[g] + s] = g * ns + s for s < cnt[g]: the dense (group, slot) row a compact token came from.
// roww (optional): the multiplicity of the compact token -- slot 0 stands for itself and the ns - cnt repeats
__global__ __launch_bounds__(256) void ragged_rowmap_kernel(const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                                                            int32_t* __restrict__ rowmap, float* __restrict__ roww, int64_t total,
                                                            int ns) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (group, slot)
    if (e >= total) return;
    const int g = (int)(e / ns), s = (int)(e % ns);
    const int c = cnt[g];
    if (s < c) {
        rowmap[off[g] + s] = (int32_t)e;
        if (roww) roww[off[g] + s] = s == 0 ? (float)(ns - c + 1) : 1.f;
    }
}

// Token assembly (csrc/assemble.hip) writing COMPACT rows: thread = (compact token u, 4-channel column).
__global__ __launch_bounds__(256) void assemble_ragged_fwd_kernel(const float* __restrict__ rppe, const float* __restrict__ dscale,
                                                                  const float* __restrict__ feats, const int* __restrict__ idx,
                                                                  const float* __restrict__ glob, const int32_t* __restrict__ rowmap,
                                                                  const int32_t* __restrict__ off, float* __restrict__ out, int n,
                                                                  int m, int ns, int c4, int groups, int rppe_compact) {
    const int64_t total = (int64_t)off[groups] * c4;             // U is read on the device: the grid covers a host bound
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % c4);
    const int64_t u = e / c4;
    const int64_t tok = rowmap[u];
    const int64_t bm = tok / ns;
    const int64_t b = bm / m;
    const float4 r = reinterpret_cast<const float4*>(rppe)[(rppe_compact ? u : tok) * c4 + c];
    const float4 f = reinterpret_cast<const float4*>(feats)[((size_t)b * n + idx[tok]) * c4 + c];
    const float4 g = reinterpret_cast<const float4*>(glob)[bm * c4 + c];
    const float d = dscale[tok];
    float4* o = reinterpret_cast<float4*>(out) + (size_t)u * 4 * c4 + c;
    o[0] = r;
    o[c4] = make_float4(f.x * d, f.y * d, f.z * d, f.w * d);
    o[2 * c4] = f;
    o[3 * c4] = g;
}

// Backward: thread = (centre, 4-channel column) walks the centre's cnt compact tokens.  d_rppe / d_dscale are the
// DENSE (B,M,ns,.) gradients the (dense) position MLP and DensityNet backward read: the compact token's gradient
// goes to its own slot, the slots of the repeats get zero (the total over the copies of a token is what counts).
template <int c4>
__global__ __launch_bounds__(256) void assemble_ragged_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ dscale,
                                                                  const float* __restrict__ feats, const int* __restrict__ idx,
                                                                  const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                                                                  float* __restrict__ d_rppe, float* __restrict__ d_dscale,
                                                                  float* __restrict__ d_feats, float* __restrict__ d_glob, int n,
                                                                  int m, int ns, int64_t total, int rppe_compact) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (centre, column)
    const bool live = e < total;
    const int64_t ec = live ? e : total - 1;                     // dead lanes shadow the last element (shuffles stay uniform)
    const int c = (int)(ec % c4);
    const int64_t bm = ec / c4;
    const int64_t b = bm / m;
    const int64_t tok0 = bm * ns;
    const int n_tok = cnt[bm];
    const int64_t u0 = off[bm];
    const float4* frows = reinterpret_cast<const float4*>(feats) + (size_t)b * n * c4 + c;
    float4 acc_g = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < ns; ++t) {
        const int64_t tok = tok0 + t;
        if (t >= n_tok) {      // uniform over the c4 lanes of a centre
            if (live) {
                if (!rppe_compact) reinterpret_cast<float4*>(d_rppe)[(size_t)tok * c4 + c] = zero;
                if (c == 0) d_dscale[tok] = 0.f;
            }
            continue;
        }
        const float4* g4 = reinterpret_cast<const float4*>(dx) + (size_t)(u0 + t) * 4 * c4 + c;
        const float4 g_r = g4[0], g_fd = g4[c4], g_f = g4[2 * c4], g_g = g4[3 * c4];
        const int row = idx[tok];
        const float4 f = frows[(size_t)row * c4];
        const float d = dscale[tok];
        float part = (g_fd.x * f.x + g_fd.y * f.y) + (g_fd.z * f.z + g_fd.w * f.w);
#pragma unroll
        for (int o = c4 >> 1; o >= 1; o >>= 1) part += __shfl_xor(part, o);
        acc_g.x += g_g.x; acc_g.y += g_g.y; acc_g.z += g_g.z; acc_g.w += g_g.w;
        if (live) {
            reinterpret_cast<float4*>(d_rppe)[(size_t)(rppe_compact ? u0 + t : tok) * c4 + c] = g_r;
            if (c == 0) d_dscale[tok] = part;
            float* df = d_feats + (((size_t)b * n + row) * c4 + c) * 4;
            atomicAdd(df + 0, g_fd.x * d + g_f.x); atomicAdd(df + 1, g_fd.y * d + g_f.y);
            atomicAdd(df + 2, g_fd.z * d + g_f.z); atomicAdd(df + 3, g_fd.w * d + g_f.w);
        }
    }
    if (live) reinterpret_cast<float4*>(d_glob)[bm * c4 + c] = acc_g;
}

// The same backward with the per-token work spread over (compact token, column) threads -- the form above gives a thread the
// 1..ns tokens of ONE centre, a chain of dependent loads per token with two waves per SIMD on the ONCE scales (0.16 ms on
// the 63 511-token scale) -- and the per-centre part (d_glob, the zeros on the repeat slots) left to (centre, column) threads.
// d_rppe / d_dscale / d_glob are the same values bit for bit; d_feats is a float-atomic sum either way.
template <int c4>
__global__ __launch_bounds__(256) void assemble_ragged_bwd_token_kernel(const float* __restrict__ dx, const float* __restrict__ dscale,
                                                                        const float* __restrict__ feats, const int* __restrict__ idx,
                                                                        const int32_t* __restrict__ rowmap, const int32_t* __restrict__ off,
                                                                        float* __restrict__ d_rppe, float* __restrict__ d_dscale,
                                                                        float* __restrict__ d_feats, int n, int m, int ns, int groups,
                                                                        int rppe_compact) {
    const int64_t total = (int64_t)off[groups] * c4;             // U is read on the device: the grid covers a host bound
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (compact token, column)
    const bool live = e < total;
    const int64_t ec = live ? e : (total > 0 ? total - 1 : 0);   // dead lanes shadow the last element (shuffles stay uniform)
    if (total == 0) return;
    const int c = (int)(ec % c4);
    const int64_t u = ec / c4;
    const int64_t tok = rowmap[u];
    const int64_t b = (tok / ns) / m;
    const float4* g4 = reinterpret_cast<const float4*>(dx) + (size_t)u * 4 * c4 + c;
    const float4 g_r = g4[0], g_fd = g4[c4], g_f = g4[2 * c4];
    const int row = idx[tok];
    const float4 f = (reinterpret_cast<const float4*>(feats) + (size_t)b * n * c4 + c)[(size_t)row * c4];
    const float d = dscale[tok];
    float part = (g_fd.x * f.x + g_fd.y * f.y) + (g_fd.z * f.z + g_fd.w * f.w);
#pragma unroll
    for (int o = c4 >> 1; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    if (!live) return;
    reinterpret_cast<float4*>(d_rppe)[(size_t)(rppe_compact ? u : tok) * c4 + c] = g_r;
    if (c == 0) d_dscale[tok] = part;
    float* df = d_feats + (((size_t)b * n + row) * c4 + c) * 4;
    atomicAdd(df + 0, g_fd.x * d + g_f.x); atomicAdd(df + 1, g_fd.y * d + g_f.y);
    atomicAdd(df + 2, g_fd.z * d + g_f.z); atomicAdd(df + 3, g_fd.w * d + g_f.w);
}

__global__ __launch_bounds__(256) void assemble_ragged_bwd_group_kernel(const float* __restrict__ dx, const int32_t* __restrict__ cnt,
                                                                        const int32_t* __restrict__ off, float* __restrict__ d_rppe,
                                                                        float* __restrict__ d_dscale, float* __restrict__ d_glob, int c4,
                                                                        int ns, int64_t total, int rppe_compact) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (centre, column)
    if (e >= total) return;
    const int c = (int)(e % c4);
    const int64_t bm = e / c4;
    const int n_tok = cnt[bm];
    const float4* g4 = reinterpret_cast<const float4*>(dx) + (size_t)off[bm] * 4 * c4 + 3 * c4 + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int t = 0;
    for (; t + 4 <= n_tok; t += 4) {                              // four independent loads in flight; the sum keeps the token order
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = g4[(size_t)(t + q) * 4 * c4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    }
    for (; t < n_tok; ++t) { const float4 v = g4[(size_t)t * 4 * c4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    reinterpret_cast<float4*>(d_glob)[bm * c4 + c] = acc;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (t = n_tok; t < ns; ++t) {                                // the repeats' slots of the dense gradients
        if (!rppe_compact) reinterpret_cast<float4*>(d_rppe)[(size_t)(bm * ns + t) * c4 + c] = zero;
        if (c == 0) d_dscale[bm * ns + t] = 0.f;
    }
}

// out (G, D) = max over the cnt tokens of a group of a + b (compact rows), arg = slot of the first maximum.
__global__ __launch_bounds__(256) void add_max_pool_ragged_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                  const int32_t* __restrict__ cnt, const int32_t* __restrict__ off,
                                                                  float* __restrict__ out, uint8_t* __restrict__ arg, int64_t groups,
                                                                  int d4) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (group, 4-channel column)
    if (e >= groups * d4) return;
    const int64_t g = e / d4;
    const int c = (int)(e % d4);
    const int s = cnt[g];
    const float4* pa = reinterpret_cast<const float4*>(a) + (size_t)off[g] * d4 + c;
    const float4* pb = reinterpret_cast<const float4*>(b) + (size_t)off[g] * d4 + c;
    float4 best = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
    uchar4 bi = make_uchar4(0, 0, 0, 0);
    for (int t = 0; t < s; ++t) {
        const float4 x = pa[(size_t)t * d4], y = pb[(size_t)t * d4];
        const float4 v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        if (v.x > best.x) { best.x = v.x; bi.x = (uint8_t)t; }
        if (v.y > best.y) { best.y = v.y; bi.y = (uint8_t)t; }
        if (v.z > best.z) { best.z = v.z; bi.z = (uint8_t)t; }
        if (v.w > best.w) { best.w = v.w; bi.w = (uint8_t)t; }
    }
    reinterpret_cast<float4*>(out)[e] = best;
    reinterpret_cast<uchar4*>(arg)[e] = bi;
}

// dx (U, D): grad_out routed to the arg-max token of each (group, channel), zeros elsewhere.  thread = (compact token, column)
__global__ __launch_bounds__(256) void max_pool_scatter_ragged_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ arg,
                                                                      const int32_t* __restrict__ rowmap, const int32_t* __restrict__ off,
                                                                      float* __restrict__ dx, int groups, int ns, int d4) {
    const int64_t total = (int64_t)off[groups] * d4;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % d4);
    const int64_t u = e / d4;
    const int32_t tok = rowmap[u];
    const int64_t g = tok / ns;
    const int t = tok % ns;
    const float4 v = reinterpret_cast<const float4*>(dout)[g * d4 + c];
    const uchar4 k = reinterpret_cast<const uchar4*>(arg)[g * d4 + c];
    reinterpret_cast<float4*>(dx)[e] = make_float4(k.x == t ? v.x : 0.f, k.y == t ? v.y : 0.f, k.z == t ? v.z : 0.f, k.w == t ? v.w : 0.f);
}

static bool ragged_c_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128 || c == 256; }

}  // namespace pda

PDA_API int pda_ragged_plan(const int32_t* idx, int32_t* cnt, int32_t* off, int32_t* rowmap, float* row_weight, int64_t groups,
                            int nsample, pda_stream_t stream) {
    PDA_REQUIRE(groups >= 0 && groups <= (1 << 24) && nsample >= 1 && nsample <= 255 && groups * nsample < INT32_MAX,
                "pda_ragged_plan: groups=%lld nsample=%d", (long long)groups, nsample);
    PDA_REQUIRE(off != nullptr, "pda_ragged_plan: null pointer");
    PDA_REQUIRE(groups == 0 || (idx && cnt && rowmap), "pda_ragged_plan: null pointer");
    if (groups > 0)
        hipLaunchKernelGGL(pda::ragged_count_kernel, dim3((unsigned)pda::divup64(groups, 256)), dim3(256), 0, (hipStream_t)stream, idx, cnt,
                           (int)groups, nsample);
    hipLaunchKernelGGL(pda::ragged_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cnt, off, (int)groups);
    if (groups > 0) {
        const int64_t total = groups * nsample;
        hipLaunchKernelGGL(pda::ragged_rowmap_kernel, dim3((unsigned)pda::divup64(total, 256)), dim3(256), 0, (hipStream_t)stream, cnt,
                           off, rowmap, row_weight, total, nsample);
    }
    return pda::check_launch("pda_ragged_plan");
}

PDA_API int pda_assemble_tokens_ragged(const float* rppe, const float* dscale, const float* feats, const int32_t* idx,
                                       const float* glob, const int32_t* rowmap, const int32_t* off, float* out, int64_t max_tokens,
                                       int b, int n, int m, int nsample, int c, int rppe_compact, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 1 && max_tokens >= 0 && pda::ragged_c_ok(c),
                "pda_assemble_tokens_ragged: b=%d n=%d m=%d nsample=%d C=%d", b, n, m, nsample, c);
    const int64_t total = max_tokens * (c / 4);
    if (total == 0 || (int64_t)b * m == 0) return PDA_OK;
    PDA_REQUIRE(rppe && dscale && feats && idx && glob && rowmap && off && out, "pda_assemble_tokens_ragged: null pointer");
    PDA_REQUIRE((((uintptr_t)rppe | (uintptr_t)feats | (uintptr_t)glob | (uintptr_t)out) & 15) == 0, "pda_assemble_tokens_ragged: alignment");
    hipLaunchKernelGGL(pda::assemble_ragged_fwd_kernel, dim3((unsigned)pda::divup64(total, 256)), dim3(256), 0, (hipStream_t)stream, rppe,
                       dscale, feats, idx, glob, rowmap, off, out, n, m, nsample, c / 4, b * m, rppe_compact);
    return pda::check_launch("pda_assemble_tokens_ragged");
}

PDA_API int pda_assemble_tokens_ragged_grad(const float* grad_out, const float* dscale, const float* feats, const int32_t* idx,
                                            const int32_t* cnt, const int32_t* off, const int32_t* rowmap, float* grad_rppe,
                                            float* grad_dscale, float* grad_feats, float* grad_glob, int64_t max_tokens, int b, int n,
                                            int m, int nsample, int c, int rppe_compact, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 1 && max_tokens >= 0 && pda::ragged_c_ok(c),
                "pda_assemble_tokens_ragged_grad: b=%d n=%d m=%d nsample=%d C=%d", b, n, m, nsample, c);
    const int64_t centre_cols = (int64_t)b * m * (c / 4);
    if (centre_cols == 0) return PDA_OK;
    PDA_REQUIRE(grad_out && dscale && feats && idx && cnt && off && grad_rppe && grad_dscale && grad_feats && grad_glob,
                "pda_assemble_tokens_ragged_grad: null pointer");
    PDA_REQUIRE((((uintptr_t)grad_out | (uintptr_t)feats | (uintptr_t)grad_rppe | (uintptr_t)grad_glob) & 15) == 0,
                "pda_assemble_tokens_ragged_grad: alignment");
    const dim3 grid((unsigned)pda::divup64(centre_cols, 256)), block(256);
    if (rowmap) {
        // per-token work over (compact token, column) threads, per-centre work over (centre, column) threads
        const int64_t tok_cols = max_tokens * (c / 4);
        if (tok_cols > 0) {
            const dim3 tgrid((unsigned)pda::divup64(tok_cols, 256));
#define PDA_ASM_TOK(C4)                                                                                                               \
    case C4: hipLaunchKernelGGL(pda::assemble_ragged_bwd_token_kernel<C4>, tgrid, block, 0, (hipStream_t)stream, grad_out, dscale, feats, \
                                idx, rowmap, off, grad_rppe, grad_dscale, grad_feats, n, m, nsample, b * m, rppe_compact); break
            switch (c / 4) { PDA_ASM_TOK(4); PDA_ASM_TOK(8); PDA_ASM_TOK(16); PDA_ASM_TOK(32); PDA_ASM_TOK(64); }
#undef PDA_ASM_TOK
        }
        hipLaunchKernelGGL(pda::assemble_ragged_bwd_group_kernel, grid, block, 0, (hipStream_t)stream, grad_out, cnt, off, grad_rppe,
                           grad_dscale, grad_glob, c / 4, nsample, centre_cols, rppe_compact);
        return pda::check_launch("pda_assemble_tokens_ragged_grad");
    }
#define PDA_ASM_BWD(C4)                                                                                                          \
    case C4: hipLaunchKernelGGL(pda::assemble_ragged_bwd_kernel<C4>, grid, block, 0, (hipStream_t)stream, grad_out, dscale, feats, idx, \
                                cnt, off, grad_rppe, grad_dscale, grad_feats, grad_glob, n, m, nsample, centre_cols, rppe_compact); break
    switch (c / 4) { PDA_ASM_BWD(4); PDA_ASM_BWD(8); PDA_ASM_BWD(16); PDA_ASM_BWD(32); PDA_ASM_BWD(64); }
#undef PDA_ASM_BWD
    return pda::check_launch("pda_assemble_tokens_ragged_grad");
}

PDA_API int pda_add_max_pool_ragged(const float* a, const float* b, const int32_t* cnt, const int32_t* off, float* out, uint8_t* arg,
                                    int64_t groups, int d, pda_stream_t stream) {
    PDA_REQUIRE(groups >= 0 && d >= 4 && (d & 3) == 0, "pda_add_max_pool_ragged: groups=%lld d=%d", (long long)groups, d);
    if (groups == 0) return PDA_OK;
    PDA_REQUIRE(a && b && cnt && off && out && arg, "pda_add_max_pool_ragged: null pointer");
    PDA_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0 && ((uintptr_t)arg & 3) == 0, "pda_add_max_pool_ragged: alignment");
    const int64_t n = groups * (d / 4);
    hipLaunchKernelGGL(pda::add_max_pool_ragged_kernel, dim3((unsigned)pda::divup64(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, cnt,
                       off, out, arg, groups, d / 4);
    return pda::check_launch("pda_add_max_pool_ragged");
}

PDA_API int pda_max_pool_scatter_ragged(const float* grad_out, const uint8_t* arg, const int32_t* rowmap, const int32_t* off,
                                        float* grad_x, int64_t max_tokens, int64_t groups, int nsample, int d, pda_stream_t stream) {
    PDA_REQUIRE(groups >= 0 && groups < INT32_MAX && max_tokens >= 0 && nsample >= 1 && d >= 4 && (d & 3) == 0,
                "pda_max_pool_scatter_ragged: groups=%lld d=%d", (long long)groups, d);
    if (groups == 0 || max_tokens == 0) return PDA_OK;
    PDA_REQUIRE(grad_out && arg && rowmap && off && grad_x, "pda_max_pool_scatter_ragged: null pointer");
    PDA_REQUIRE((((uintptr_t)grad_out | (uintptr_t)grad_x) & 15) == 0 && ((uintptr_t)arg & 3) == 0, "pda_max_pool_scatter_ragged: alignment");
    const int64_t n = max_tokens * (d / 4);
    hipLaunchKernelGGL(pda::max_pool_scatter_ragged_kernel, dim3((unsigned)pda::divup64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, arg, rowmap, off, grad_x, (int)groups, nsample, d / 4);
    return pda::check_launch("pda_max_pool_scatter_ragged");
}
