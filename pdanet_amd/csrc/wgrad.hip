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
#include "split_bf16.h"

#include <stdlib.h>

namespace pda {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_KC = 32;       // token rows per LDS chunk
constexpr int WG_TILE = 128;    // output tile edge

// global_load_lds_dwordx4 as inline asm: with the builtin hipcc (ROCm 7.2) drains vmcnt(0) before the first
// LDS read that follows, i.e. before the MFMAs the transfer is meant to overlap.  The caller retires the
// transfers with an explicit s_waitcnt vmcnt(0) before the barrier that precedes the reads of that buffer.
__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_dst_wave_uniform) {
    uint32_t keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_wave_uniform);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

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

    // Full 128-column tiles take their chunks by LDS-DMA (global_load_lds_dwordx4: one wave instruction lands two
    // whole 512-byte rows, no VGPR staging, no ds_write pass); the chunk after the one being multiplied is in
    // flight under the MFMAs.  Ragged tiles / the ragged last chunk of a slice go through registers (zero fill).
    const bool full_tile = (n0 + WG_TILE <= N) && (m0 + WG_TILE <= M);
    auto dma_chunk = [&](int64_t t0, int b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = w * 8 + 2 * i;                                   // this wave instruction: rows row, row + 1
            const int64_t t = t0 + row + (lane >> 5);
            const int col = (lane & 31) * 4;
            lds_dma16(G + t * N + n0 + col, &ldsG[b][row][0]);
            lds_dma16(X + t * M + m0 + col, &ldsX[b][row][0]);
        }
    };
    auto compute = [&](int buf) {
        // operand fragments one k-pair ahead of the MFMAs that use them (LDS latency under the MFMA chain)
        float fa[2][2], fb[2][2];
        fa[0][0] = ldsG[buf][h][wn * 64 + c]; fa[0][1] = ldsG[buf][h][wn * 64 + 32 + c];
        fb[0][0] = ldsX[buf][h][wm * 64 + c]; fb[0][1] = ldsX[buf][h][wm * 64 + 32 + c];
#pragma unroll
        for (int kk = 0; kk < WG_KC / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < WG_KC / 2) {
                const int k = 2 * (kk + 1) + h;
                fa[nxt][0] = ldsG[buf][k][wn * 64 + c]; fa[nxt][1] = ldsG[buf][k][wn * 64 + 32 + c];
                fb[nxt][0] = ldsX[buf][k][wm * 64 + c]; fb[nxt][1] = ldsX[buf][k][wm * 64 + 32 + c];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
            bsum[0] += fa[cur][0]; bsum[1] += fa[cur][1];
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int buf = 0;
    int64_t t0 = t_begin;
    if (full_tile) {
        // pure LDS-DMA loop over the whole chunks (kept free of register-staged loads: hipcc drains vmcnt(0) at
        // any use of an ordinary load's result while a DMA is in flight)
        const int64_t n_full = (t_end - t_begin) / WG_KC;
        if (n_full > 0) {
            dma_chunk(t_begin, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int64_t i = 0; i < n_full; ++i) {
                if (i + 1 < n_full) dma_chunk(t_begin + (i + 1) * WG_KC, buf ^ 1);
                compute(buf);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next chunk has landed
                __syncthreads();
                buf ^= 1;
            }
            t0 = t_begin + n_full * WG_KC;
        }
    }
    if (t0 < t_end) {                  // ragged tiles, and the ragged last chunk of a slice: through registers
        load_chunk(t0);
        store_chunk(buf);
        __syncthreads();
        for (; t0 < t_end; t0 += WG_KC) {
            const bool more = t0 + WG_KC < t_end;
            if (more) load_chunk(t0 + WG_KC);
            compute(buf);
            if (more) store_chunk(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
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

// ---- wide layers over many tokens: the same reduction on the bf16 matrix cores (split-bf16, csrc/split_bf16.h) ------
// wgrad_kernel is bound by v_mfma_f32_32x32x2_f32 (157 TFLOP/s: it reaches 100-127).  Here both operands are split into
// three bf16 terms and six of the nine products are kept (gemm_split.hip has the error analysis): 2.67x the f32 matrix
// rate at f32 accuracy.  A workgroup (8 waves, 4 x 2) owns a 256 x 256 tile of dW and one slice of T and works in steps
// of 16 tokens.  The MFMA operands want 8 consecutive tokens of one column per lane, the matrices are [token][column]:
//   * producer role: wave w owns columns 32 w .. 32 w + 31 of both operands.  Lane (c, h) loads the eight values
//     [8 h + j][c] of a step straight into registers (128 contiguous bytes per half-wave and row), two steps ahead,
//     splits them (each value is split ONCE per workgroup) and writes the three planes as ready-made MFMA fragments
//     [operand][32-column block][plane][lane] x 16 bytes into one of two LDS buffers -- no transposing store, no bank
//     conflict on either side;
//   * consumer role: wave (wn, wm) computes 64 (n) x 128 (m): 18 ds_read_b128 for 48 MFMAs per step.
// The split and the plane stores of step s + 1 sit under the MFMAs of step s; one barrier per step.
// Timing experiments only (results are wrong): -DWSP_ABL=1 no split arithmetic, 2 one MFMA in twelve, 3 no loads in the loop,
// 4 no barrier.  rocprofv3 counters on 131072 x 512 x 512: the bf16 pipe is 69 % busy at a 1.8 GHz clock = 1.3 PFLOP/s of
// bf16 MFMA work on random data, which is where the chip holds its clock down (MI355X_MICROARCH.md, DVFS give-back):
// 7 % fewer wave cycles from the deferred quarter came back as a 4 % lower clock.
#ifndef WSP_ABL
#define WSP_ABL 0
#endif
constexpr int WSP_TILE_U4 = 2 * 8 * 3 * 64;      // uint4 per plane buffer: 48 KB

// XBN: X is the PRE-BatchNorm tensor of the layer; its training-mode BatchNorm + ReLU (bn_apply_kernel's expression,
// csrc/bn_relu.hip) is applied to the eight values of a lane's column right before they are split -- the layer's activation
// is never read from (or written to) HBM.  Rows behind the slice read as 0 in both operands, so what the transform makes
// of a zero row of X meets a zero row of G.
template <bool XBN>
__global__ __launch_bounds__(512, 1)
void wgrad_split_kernel(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ part_w,
                        float* __restrict__ part_b, int64_t T, int M, int N, int tiles_m, int tiles_n, int S, int64_t KS,
                        int nblocks, const float* __restrict__ x_mi, const float* __restrict__ x_g, const float* __restrict__ x_b) {
    extern __shared__ uint4 wsp_planes[];        // 2 * WSP_TILE_U4
    const int per_xcd = (nblocks + 7) / 8;       // workgroups of one slice on one XCD (they read the same rows)
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= nblocks) return;
    const int tiles = tiles_m * tiles_n;
    const int tile = logical % tiles, s = logical / tiles;
    const int n0 = (tile / tiles_m) * 256, m0 = (tile % tiles_m) * 256;
    const int64_t t_begin = (int64_t)s * KS, t_end = (t_begin + KS < T) ? t_begin + KS : T;
    const int len = (int)(t_end - t_begin), ksteps = (len + 15) >> 4;

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int wn = w >> 1, wm = w & 1;
    // producer addresses: a buffer resource over the wave's 32 columns of the slice (wave-uniform), one per-lane offset
    // and the row as the instruction's scalar offset.  The resource ends behind the last row of the slice: rows beyond
    // it (the ragged last step, the two look-ahead steps) read as 0.0 by the hardware's range check.
    const uint32_t span_g = (uint32_t)((((int64_t)len - 1) * N + 32) * 4), span_x = (uint32_t)((((int64_t)len - 1) * M + 32) * 4);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(G + t_begin * N + n0 + 32 * w), 0, span_g, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(X + t_begin * M + m0 + 32 * w), 0, span_x, 0x00020000);
    const int vg = (8 * h * N + c) * 4, vx = (8 * h * M + c) * 4;
    auto load1 = [&](int st, int j, float (&g)[8], float (&x)[8]) {
        g[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, vg, (16 * st + j) * N * 4, 0));
        x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vx, (16 * st + j) * M * 4, 0));
    };
    auto load = [&](int st, float (&g)[8], float (&x)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) load1(st, j, g, x);
    };
    float xmu = 0.f, xis = 0.f, xga = 0.f, xbe = 0.f;       // XBN: the constants of this lane's column of X
    if constexpr (XBN) { const int m = m0 + 32 * w + c; xmu = x_mi[m]; xis = x_mi[M + m]; xga = x_g[m]; xbe = x_b[m]; }
    auto xt = [&](float v) { return XBN ? fmaxf((v - xmu) * xis * xga + xbe, 0.f) : v; };
    float bsum = 0.f;
    const bool do_bias = part_b && m0 == 0;
    gs_f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int aoff = (2 * wn) * 192 + lane, boff = (8 + 4 * wm) * 192 + lane;
    uint4* const my_planes = wsp_planes + w * 192 + lane;          // + b * WSP_TILE_U4 [+ 8 * 192 for X] + plane * 64

    // One step.  ga / xa hold the raw values of step st + 1 (split here), gb / xb receive those of step st + 2.  The
    // order below is the issue order (sched_barrier fences):
    //   * the last quarter of the PREVIOUS step's MFMAs (column block 3: fragments An / Bf[1] still in registers) is issued
    //     behind this step's barrier, over the latency of this step's first fragment reads -- the two waves of a SIMD reach
    //     the barrier together, so nothing else could keep the matrix pipe busy there;
    //   * the split's VALU work in eight pieces, each behind three MFMAs that are independent of it;
    //   * the sixteen loads two per three MFMAs (all sixteen at once held up both waves of a SIMD).
    gs_bf16x8 Bf[2][3];
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // the six products, smallest terms first
    auto step = [&](int st, const float (&ga)[8], const float (&xa)[8], float (&gb)[8], float (&xb)[8],
                    const gs_bf16x8 (&Ap)[2][3], gs_bf16x8 (&An)[2][3]) {
#if WSP_ABL != 4
        __syncthreads();                   // planes of step st are complete; everyone is done with step st - 1
#endif
        const uint4* buf = wsp_planes + (st & 1) * WSP_TILE_U4;
        uint4* out = my_planes + ((st + 1) & 1) * WSP_TILE_U4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) An[i][pl] = __builtin_bit_cast(gs_bf16x8, buf[aoff + i * 192 + pl * 64]);
        auto read_b = [&](int jt) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) Bf[jt & 1][pl] = __builtin_bit_cast(gs_bf16x8, buf[boff + jt * 192 + pl * 64]);
        };
        auto mfma3 = [&](const gs_bf16x8 (&A)[2][3], int jt, int k) {     // MFMAs 3k .. 3k + 2 of the 12 of column block jt
#pragma unroll
            for (int e = 3 * k; e < 3 * k + 3; ++e) {
                const int i = e & 1, p = e >> 1;                           // alternate the two accumulators
#if WSP_ABL == 2
                if (e != 0) continue;
#endif
                acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][PA[p]], Bf[jt & 1][PB[p]], acc[i][jt], 0, 0, 0);
            }
        };
        read_b(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) mfma3(Ap, 3, q);                       // the previous step's column block 3
        __builtin_amdgcn_sched_barrier(0);
        read_b(1);
        uint32_t ph[4], pm[4], pl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mfma3(An, 0, q);
            __builtin_amdgcn_sched_barrier(0);
#if WSP_ABL == 1
            ph[q] = __float_as_uint(ga[2 * q]); pm[q] = __float_as_uint(ga[2 * q + 1]); pl[q] = ph[q] ^ pm[q];
#else
            split2(ga[2 * q], ga[2 * q + 1], ph[q], pm[q], pl[q]);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        out[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        out[64] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        out[128] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        bsum += ((ga[0] + ga[1]) + (ga[2] + ga[3])) + ((ga[4] + ga[5]) + (ga[6] + ga[7]));      // used by the m0 == 0 workgroups
        read_b(2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mfma3(An, 1, q);
            __builtin_amdgcn_sched_barrier(0);
#if WSP_ABL == 1
            ph[q] = __float_as_uint(xa[2 * q]); pm[q] = __float_as_uint(xa[2 * q + 1]); pl[q] = ph[q] ^ pm[q];
#else
            split2(xt(xa[2 * q]), xt(xa[2 * q + 1]), ph[q], pm[q], pl[q]);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        out[8 * 192] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        out[8 * 192 + 64] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        out[8 * 192 + 128] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        read_b(3);                          // stays in Bf[1] for the MFMAs behind the next barrier
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mfma3(An, 2, q);
#if WSP_ABL != 3
            __builtin_amdgcn_sched_barrier(0);
            load1(st + 2, 2 * q, gb, xb);
            load1(st + 2, 2 * q + 1, gb, xb);
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    auto split_store = [&](const float (&g)[8], const float (&x)[8], int b) {     // the first step's planes
        uint4* out = my_planes + b * WSP_TILE_U4;
        uint32_t ph[4], pm[4], pl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split2(g[2 * q], g[2 * q + 1], ph[q], pm[q], pl[q]);
        out[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        out[64] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        out[128] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) split2(xt(x[2 * q]), xt(x[2 * q + 1]), ph[q], pm[q], pl[q]);
        out[8 * 192] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        out[8 * 192 + 64] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        out[8 * 192 + 128] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        bsum += ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
    };
    float g1[8], x1[8], g2[8], x2[8];
    gs_bf16x8 A0[2][3], A1[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            A0[i][pl] = __builtin_bit_cast(gs_bf16x8, make_uint4(0u, 0u, 0u, 0u));      // "the step before the first": zeros
            Bf[1][pl] = A0[i][pl];
        }
    load(0, g1, x1);
    split_store(g1, x1, 0);
    load(1, g1, x1);
    for (int st = 0; st < ksteps; st += 2) {       // in pairs (the register sets swap roles); a step behind the last adds zeros
        step(st, g1, x1, g2, x2, A0, A1);
        step(st + 1, g2, x2, g1, x1, A1, A0);
    }
#pragma unroll
    for (int e = 0; e < 12; ++e)                   // column block 3 of the last step
        acc[e & 1][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[e & 1][PA[e >> 1]], Bf[1][PB[e >> 1]], acc[e & 1][3], 0, 0, 0);
    // D[row n][col m]: lane (c = m, h), register r -> n = (r & 3) + 8 * (r >> 2) + 4 * h
    float* pw = part_w + (size_t)s * N * M;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int m = m0 + 128 * wm + 32 * jt + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 64 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                pw[(size_t)n * M + m] = acc[i][jt][r];
            }
        }
    if (do_bias) {
        const float v = bsum + __shfl_xor(bsum, 32);
        if (h == 0) part_b[(size_t)s * N + n0 + 32 * w + c] = v;
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

// ---- narrow layers (M, N <= 64): an HBM-bound streaming reduction --------------------------------------------------
// The set-abstraction MLPs of layer 0 (4 -> 16 -> 16 -> 32, 4 -> 32 -> 32 -> 64 over 0.5-1 M tokens) and the position MLPs
// (12 -> 32 -> 64) have weight gradients with a 16..64-wide output and a reduction over up to a million tokens; the
// library runs them 3-7x off the time their 40-400 MB of operands need to stream (0.1-0.2 ms each, 1.4 ms per step).
// Here a wave streams token pairs straight into the operands of v_mfma_f32_32x32x2_f32 (lane = column, lane half = the
// token of the pair: 128 contiguous bytes per half-wave and row), eight pairs in flight, and keeps the whole N x M product
// in NT x MT accumulator tiles; the four waves of a workgroup are summed through LDS, the workgroups by the fixed-order
// second stage above.  Columns beyond M / N load zeros.
constexpr int WS_UNROLL = 8;
constexpr int WS_MAX_SLICES = 1024;

template <int NT, int MT>
__global__ __launch_bounds__(256) void wgrad_skinny_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                           float* __restrict__ part_w, float* __restrict__ part_b, int64_t T, int M,
                                                           int N, int64_t KS) {
    __shared__ float red[NT * MT * 1024 + 64 * NT];
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int64_t t_begin = (int64_t)blockIdx.x * KS, t_end = (t_begin + KS < T) ? t_begin + KS : T;
    f32x16 acc[NT][MT];
    float sb[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        sb[a] = 0.f;
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }
    bool gcol[NT], xcol[MT];
#pragma unroll
    for (int a = 0; a < NT; ++a) gcol[a] = a * 32 + i < N;
#pragma unroll
    for (int b = 0; b < MT; ++b) xcol[b] = b * 32 + i < M;
    // wave w takes the pairs w, w + 4, ... of the slice, WS_UNROLL pairs per trip
    for (int64_t t0 = t_begin + 2 * w; t0 < t_end; t0 += 8 * WS_UNROLL) {
        float ga[WS_UNROLL][NT], xb[WS_UNROLL][MT];
#pragma unroll
        for (int u = 0; u < WS_UNROLL; ++u) {
            const int64_t t = t0 + 8 * u + h;
            const bool ok = t < t_end;
#pragma unroll
            for (int a = 0; a < NT; ++a) ga[u][a] = (ok && gcol[a]) ? G[t * N + a * 32 + i] : 0.f;
#pragma unroll
            for (int b = 0; b < MT; ++b) xb[u][b] = (ok && xcol[b]) ? X[t * M + b * 32 + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < WS_UNROLL; ++u)
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                sb[a] += ga[u][a];
#pragma unroll
                for (int b = 0; b < MT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[u][a], xb[u][b], acc[a][b], 0, 0, 0);
            }
    }
    // workgroup sum in wave order (fixed): C layout of a tile: row n = (r & 3) + 8 (r >> 2) + 4 h, column m = i
    for (int turn = 0; turn < 4; ++turn) {
        if (w == turn) {
#pragma unroll
            for (int a = 0; a < NT; ++a) {
#pragma unroll
                for (int b = 0; b < MT; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* d = red + ((a * MT + b) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i;
                        *d = turn == 0 ? acc[a][b][r] : *d + acc[a][b][r];
                    }
                float* d = red + NT * MT * 1024 + a * 64 + lane;
                *d = turn == 0 ? sb[a] : *d + sb[a];
            }
        }
        __syncthreads();
    }
    float* pw = part_w + (size_t)blockIdx.x * N * M;
    for (int e = tid; e < NT * MT * 1024; e += 256) {
        const int tile = e >> 10, n = (tile / MT) * 32 + ((e >> 5) & 31), m = (tile % MT) * 32 + (e & 31);
        if (n < N && m < M) pw[n * M + m] = red[e];
    }
    if (part_b) {
        for (int n = tid; n < N; n += 256) {
            const int o = NT * MT * 1024 + (n >> 5) * 64 + (n & 31);
            part_b[(size_t)blockIdx.x * N + n] = red[o] + red[o + 32];
        }
    }
}

static bool wgrad_is_skinny(int M, int N) { return M <= 64 && N <= 64; }
static void wgrad_skinny_plan(int64_t T, int& S, int64_t& KS) {
    int64_t s = T / 512 > 0 ? T / 512 : 1;                 // at least 512 token rows (64 pairs per wave) per workgroup
    if (s > WS_MAX_SLICES) s = WS_MAX_SLICES;
    KS = divup64(divup64(T, s), 8) * 8;
    S = (int)divup64(T, KS);
}

static void wgrad_plan(int64_t T, int M, int N, int& tiles_m, int& tiles_n, int& S, int64_t& KS) {
    tiles_m = divup(M, WG_TILE); tiles_n = divup(N, WG_TILE);
    const int tiles = tiles_m * tiles_n;
    int64_t s = 512 / tiles;                       // ~2 workgroups per CU
    // few tokens: every workgroup writes 64 KB of partials whatever its slice, and 512 of them are 32 MB to write and to sum
    // -- one workgroup per CU once a slice would be under 128 rows; never under 64 rows (two chunks) per slice.  (A slice
    // of 256 rows is 14 us of f32 MFMA on its 128 x 128 tile: that floor made 8192-token layers take 27 us.)
    if (T / s < 128 && tiles <= 256) s = 256 / tiles;
    const int64_t max_s = T / 64 > 0 ? T / 64 : 1;
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    KS = divup64(divup64(T, s), WG_KC) * WG_KC;
    S = (int)divup64(T, KS);
}

// The split form: both widths whole 256-column tiles and enough work that its 256 KB of partials per workgroup
// (64 MB per call, written and summed once) stay a small part of the call.  PDA_WGRAD_SPLIT=0 keeps the f32-MFMA form,
// =2 takes the split form whenever the shape allows.
static bool wgrad_use_split(int64_t T, int M, int N) {
    static const int mode = [] { const char* e = getenv("PDA_WGRAD_SPLIT"); return e ? atoi(e) : 1; }();   // 2: whenever the shape allows
    if (mode == 0 || (M & 255) || (N & 255) || T < 4096) return false;
    return mode == 2 || (double)T * M * N >= 2.0e9;      // measured cross-over: 32768 x 256 x 256 wins, 12979 x 256 x 256 loses
}
static void wgrad_split_plan(int64_t T, int M, int N, int& tiles_m, int& tiles_n, int& S, int64_t& KS) {
    tiles_m = M / 256; tiles_n = N / 256;
    const int tiles = tiles_m * tiles_n;
    int64_t s = 256 / tiles;                       // one workgroup per CU
    const int64_t max_s = T / 256 > 0 ? T / 256 : 1;
    if (s > max_s) s = max_s;
    const int64_t min_s = divup64(T * (M > N ? M : N) * 4, (int64_t)1 << 31);     // a slice is addressed with 32-bit offsets
    if (s < min_s) s = min_s;
    if (s < 1) s = 1;
    KS = divup64(divup64(T, s), 16) * 16;
    S = (int)divup64(T, KS);
}

}  // namespace pda

PDA_API int64_t pda_linear_wgrad_scratch_bytes(int64_t tokens, int in_features, int out_features) {
    if (tokens <= 0 || in_features <= 0 || out_features <= 0) return 0;
    int tm, tn, S;
    int64_t KS;
    if (pda::wgrad_is_skinny(in_features, out_features)) pda::wgrad_skinny_plan(tokens, S, KS);
    else if (pda::wgrad_use_split(tokens, in_features, out_features)) pda::wgrad_split_plan(tokens, in_features, out_features, tm, tn, S, KS);
    else pda::wgrad_plan(tokens, in_features, out_features, tm, tn, S, KS);
    return (int64_t)S * ((int64_t)in_features * out_features + out_features) * (int64_t)sizeof(float);
}

PDA_API int pda_linear_wgrad_form(int64_t tokens, int in_features, int out_features) {
    if (pda::wgrad_is_skinny(in_features, out_features)) return 1;
    return pda::wgrad_use_split(tokens, in_features, out_features) ? 2 : 0;
}

namespace pda {
static int linear_wgrad(const float* x, const float* grad_out, float* grad_weight, float* grad_bias, void* scratch, int64_t tokens,
                        int in_features, int out_features, pda_stream_t stream, const float* x_mi, const float* x_g, const float* x_b);
}

PDA_API int pda_linear_wgrad(const float* x, const float* grad_out, float* grad_weight, float* grad_bias, void* scratch,
                             int64_t tokens, int in_features, int out_features, pda_stream_t stream) {
    return pda::linear_wgrad(x, grad_out, grad_weight, grad_bias, scratch, tokens, in_features, out_features, stream, nullptr, nullptr, nullptr);
}

// dW = dY^T relu(bn(x)): x is the PRE-BatchNorm tensor, (x_mean_invstd, x_gamma, x_beta) the layer's batch statistics and
// affine parameters.  Only the split-bf16 form has the transform in its operand path (pda_linear_wgrad_form == 2).
PDA_API int pda_linear_wgrad_bn(const float* x, const float* grad_out, float* grad_weight, float* grad_bias, void* scratch,
                                int64_t tokens, int in_features, int out_features, const float* x_mean_invstd, const float* x_gamma,
                                const float* x_beta, pda_stream_t stream) {
    PDA_REQUIRE(x_mean_invstd && x_gamma && x_beta, "pda_linear_wgrad_bn: null pointer");
    if (pda_linear_wgrad_form(tokens, in_features, out_features) != 2) {
        pda::set_error("pda_linear_wgrad_bn: tokens=%lld in=%d out=%d does not run on the split form", (long long)tokens, in_features, out_features);
        return PDA_ERR_UNSUPPORTED;
    }
    return pda::linear_wgrad(x, grad_out, grad_weight, grad_bias, scratch, tokens, in_features, out_features, stream, x_mean_invstd, x_gamma, x_beta);
}

int pda::linear_wgrad(const float* x, const float* grad_out, float* grad_weight, float* grad_bias, void* scratch, int64_t tokens,
                             int in_features, int out_features, pda_stream_t stream, const float* x_mi, const float* x_g, const float* x_b) {
    const int M = in_features, N = out_features;
    PDA_REQUIRE(tokens >= 1 && M >= 4 && N >= 4 && (M & 3) == 0 && (N & 3) == 0,
                "pda_linear_wgrad: tokens=%lld in=%d out=%d (features must be multiples of 4)", (long long)tokens, M, N);
    PDA_REQUIRE(x && grad_out && grad_weight && scratch, "pda_linear_wgrad: null pointer");
    PDA_REQUIRE((((uintptr_t)x | (uintptr_t)grad_out) & 15) == 0, "pda_linear_wgrad: x / grad_out must be 16-byte aligned");
    int tm, tn, S;
    int64_t KS;
    hipStream_t st = (hipStream_t)stream;
    if (pda::wgrad_is_skinny(M, N)) {
        pda::wgrad_skinny_plan(tokens, S, KS);
        float* pw = (float*)scratch;
        float* pb = grad_bias ? pw + (size_t)S * N * M : (float*)nullptr;
        const int nt = N > 32 ? 2 : 1, mt = M > 32 ? 2 : 1;
#define PDA_WS_CASE(A, B) hipLaunchKernelGGL((pda::wgrad_skinny_kernel<A, B>), dim3(S), dim3(256), 0, st, x, grad_out, pw, pb, tokens, M, N, KS)
        if (nt == 1 && mt == 1) PDA_WS_CASE(1, 1);
        else if (nt == 1) PDA_WS_CASE(1, 2);
        else if (mt == 1) PDA_WS_CASE(2, 1);
        else PDA_WS_CASE(2, 2);
#undef PDA_WS_CASE
        const int64_t nm = (int64_t)N * M;
        hipLaunchKernelGGL(pda::wgrad_reduce_kernel, dim3((unsigned)pda::divup64(nm, 64)), dim3(1024), 0, st, pw, pb, grad_weight,
                           grad_bias, S, nm, N);
        return pda::check_launch("pda_linear_wgrad");
    }
    if (pda::wgrad_use_split(tokens, M, N)) {
        pda::wgrad_split_plan(tokens, M, N, tm, tn, S, KS);
        const int nblocks = S * tm * tn;
        float* part_w = (float*)scratch;
        float* part_b = part_w + (size_t)S * N * M;
        static pda::PerDevice<bool> lds_ok;
        const bool ok = lds_ok.get([] {
            return hipFuncSetAttribute((const void*)pda::wgrad_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * pda::WSP_TILE_U4 * 16) == hipSuccess &&
                   hipFuncSetAttribute((const void*)pda::wgrad_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * pda::WSP_TILE_U4 * 16) == hipSuccess; });
        PDA_REQUIRE(ok, "pda_linear_wgrad: %d bytes of dynamic LDS refused", 2 * pda::WSP_TILE_U4 * 16);
        if (x_mi)
            hipLaunchKernelGGL(pda::wgrad_split_kernel<true>, dim3(pda::divup(nblocks, 8) * 8), dim3(512), 2 * pda::WSP_TILE_U4 * 16, st,
                               x, grad_out, part_w, grad_bias ? part_b : (float*)nullptr, tokens, M, N, tm, tn, S, KS, nblocks, x_mi, x_g, x_b);
        else
            hipLaunchKernelGGL(pda::wgrad_split_kernel<false>, dim3(pda::divup(nblocks, 8) * 8), dim3(512), 2 * pda::WSP_TILE_U4 * 16, st,
                               x, grad_out, part_w, grad_bias ? part_b : (float*)nullptr, tokens, M, N, tm, tn, S, KS, nblocks, x_mi, x_g, x_b);
        const int64_t nm = (int64_t)N * M;
        hipLaunchKernelGGL(pda::wgrad_reduce_kernel, dim3((unsigned)pda::divup64(nm, 64)), dim3(1024), 0, st, part_w, part_b,
                           grad_weight, grad_bias, S, nm, N);
        return pda::check_launch("pda_linear_wgrad");
    }
    pda::wgrad_plan(tokens, M, N, tm, tn, S, KS);
    const int nblocks = S * tm * tn;
    float* part_w = (float*)scratch;
    float* part_b = part_w + (size_t)S * N * M;
    hipLaunchKernelGGL(pda::wgrad_kernel, dim3(pda::divup(nblocks, 8) * 8), dim3(256), 0, st, x, grad_out, part_w,
                       grad_bias ? part_b : (float*)nullptr, tokens, M, N, tm, tn, S, KS, nblocks);
    const int64_t nm = (int64_t)N * M;
    hipLaunchKernelGGL(pda::wgrad_reduce_kernel, dim3((unsigned)pda::divup64(nm, 64)), dim3(1024), 0, st, part_w, part_b, grad_weight,
                       grad_bias, S, nm, N);
    return pda::check_launch("pda_linear_wgrad");
}

// ---- column sums of a bf16 (rows, cols) matrix: the bias gradient in dense-bf16 mode ---------------------------
// (fp32 mode gets it from wgrad_kernel on the way.)  Two deterministic stages like everything else here: per-block fp32
// partials over a strided set of rows (16-byte loads of 8 bf16 per lane), then a fixed-order double sum per column.
namespace pda {

constexpr int CS_BLOCKS = 512;

__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ g, float* __restrict__ partial, int64_t rows,
                                                          int cols, int nv, int rpb) {
    __shared__ float lds[256 * 8];
    const int tid = threadIdx.x;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int v = tid % nv, r0 = tid / nv;
    if (tid < nv * rpb) {
        for (int64_t r = (int64_t)blockIdx.x * rpb + r0; r < rows; r += (int64_t)gridDim.x * rpb) {
            const uint4 u = *reinterpret_cast<const uint4*>(g + r * cols + 8 * v);
            acc[0] += __uint_as_float(u.x << 16); acc[1] += __uint_as_float(u.x & 0xffff0000u);
            acc[2] += __uint_as_float(u.y << 16); acc[3] += __uint_as_float(u.y & 0xffff0000u);
            acc[4] += __uint_as_float(u.z << 16); acc[5] += __uint_as_float(u.z & 0xffff0000u);
            acc[6] += __uint_as_float(u.w << 16); acc[7] += __uint_as_float(u.w & 0xffff0000u);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) lds[tid * 8 + k] = acc[k];
    __syncthreads();
    if (tid < nv) {
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rpb; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += lds[(r * nv + tid) * 8 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) partial[(size_t)blockIdx.x * cols + 8 * tid + k] = s[k];
    }
}

// 16 columns x 16 slices of the per-block partials per workgroup
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblocks, int cols,
                                                           float* __restrict__ out) {
    __shared__ double red[16][16];
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double a = 0.0;
    if (c < cols) {
        float va[8];
        for (int k0 = part; k0 < nblocks; k0 += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 16 * u;
                va[u] = k < nblocks ? partial[(size_t)k * cols + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)va[u];
        }
    }
    red[part][cl] = a;
    __syncthreads();
    if (part != 0 || c >= cols) return;
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) s += red[p][cl];
    out[c] = (float)s;
}

}  // namespace pda

PDA_API int64_t pda_colsum_scratch_bytes(int cols) { return (int64_t)pda::CS_BLOCKS * (cols > 0 ? cols : 0) * (int64_t)sizeof(float); }

PDA_API int pda_colsum_bf16(const uint16_t* g, float* out, void* scratch, int64_t rows, int cols, pda_stream_t stream) {
    PDA_REQUIRE(rows >= 1 && cols >= 8 && cols <= 2048 && (cols & 7) == 0, "pda_colsum_bf16: rows=%lld cols=%d (cols: multiple of 8, <= 2048)",
                (long long)rows, cols);
    PDA_REQUIRE(g && out && scratch, "pda_colsum_bf16: null pointer");
    PDA_REQUIRE(((uintptr_t)g & 15) == 0, "pda_colsum_bf16: g must be 16-byte aligned");
    const int nv = cols / 8, rpb = 256 / nv;
    const int64_t steps = pda::divup64(rows, (int64_t)rpb * 8);
    const int grid = (int)(steps < pda::CS_BLOCKS ? steps : pda::CS_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pda::colsum_bf16_kernel, dim3(grid), dim3(256), 0, st, g, (float*)scratch, rows, cols, nv, rpb);
    hipLaunchKernelGGL(pda::colsum_final_kernel, dim3(pda::divup(cols, 16)), dim3(256), 0, st, (const float*)scratch, grid, cols, out);
    return pda::check_launch("pda_colsum_bf16");
}
