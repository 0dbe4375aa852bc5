// gemm_split.hip -- f32 GEMMs of the training step on the bf16 matrix cores, operands split into three bf16 terms.
//
// MI355X has no xf32/TF32 form and its f32-input MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate
// (157 TFLOP/s against ~2.5 PFLOP/s, MI355X_MICROARCH.md "Matrix cores").  An f32 value x is EXACTLY the sum of
// three bf16 values (8 + 8 + 8 significant bits, same exponent range):
//     h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m),      x = h + m + l,
// so a product a*b = sum of nine bf16*bf16 products, each exact in f32.  This file keeps the six largest,
//     ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh,
// and drops am*bl, al*bm, al*bl (relative size <= 2^-24 each: below the rounding of an f32 multiply).  The sums
// accumulate in the MFMA's f32 accumulator like the f32-input form's do.  Six v_mfma_f32_32x32x16_bf16 (6 x 32
// cycles) replace eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles): 2.67x the f32 matrix rate at f32 accuracy
// (tests/test_gemm_split.py measures both against an f64 product).
//
// Replaces the reference's cuDNN/cuBLAS f32 1x1 convolutions and nn.Linear calls of the group MLPs
// (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:1657-1662) in training.
//
//   lin_split_kernel<KS>:  Y (T, N) = X (T, K) W^T [+ bias] with K = 16 KS.
//     * a wave owns 32 tokens: lane (r = lane & 31, h = lane >> 5) keeps X[token r][16 s + 8 h + j], j < 8, of every
//       k-step s in registers (f32) for the whole kernel -- X is read from HBM once -- and splits the eight values of
//       a k-step into bf16 B fragments right before they are used (44 VALU instructions hidden under 24 MFMAs);
//     * the weights are split ONCE by the pack kernel into fragment-ordered bf16 planes
//       [chunk of 128 outputs][k-step][32-row block][plane h/m/l][lane][8 bf16]; a workgroup (4 waves, 128 tokens)
//       streams them through a double-buffered LDS tile (2 k-steps = 24 KB per buffer, one barrier per 48 MFMAs),
//       global -> registers -> LDS, the loads of the next tile in flight under the MFMAs of the current one.
#include "pda_common.h"
#include "split_bf16.h"

#include <stdlib.h>

namespace pda {

constexpr int GS_MAX_N = 2048;                   // widest output (bias staged in LDS)
constexpr int GS_KSTEP = 12 * 64;                // uint4 per k-step of a 128-output chunk (4 row blocks x 3 planes x 64 lanes)

// W (rows, cols) row-major (trans: the source holds W^T, (cols, rows) row-major) -> fragment-ordered planes.
__device__ __forceinline__ void split_pack_one(const float* __restrict__ w, uint32_t* __restrict__ wf, int64_t o, int rows, int cols,
                                               int KS, int trans) {
    const int pr = (int)(o & 3), lane = (int)((o >> 2) & 63);
    int64_t rest = o >> 8;
    const int rb = (int)(rest & 3); rest >>= 2;
    const int s = (int)(rest % KS), c = (int)(rest / KS);
    const int row = c * 128 + rb * 32 + (lane & 31);
    const int k = 16 * s + 8 * (lane >> 5) + 2 * pr;
    float v[2];
    for (int e = 0; e < 2; ++e)
        v[e] = (row < rows && k + e < cols) ? (trans ? w[(size_t)(k + e) * rows + row] : w[(size_t)row * cols + k + e]) : 0.f;
    uint32_t h, m, l;
    split2(v[0], v[1], h, m, l);
    const size_t base = ((((size_t)c * KS + s) * 4 + rb) * 3) * 256 + (size_t)lane * 4 + pr;    // uint32 units
    wf[base] = h; wf[base + 256] = m; wf[base + 512] = l;
}

__global__ void split_pack_kernel(const float* __restrict__ w, uint32_t* __restrict__ wf, int rows, int cols, int KS,
                                  int chunks, int trans) {
    const int64_t total = (int64_t)chunks * KS * 4 * 64 * 4;       // one thread per (chunk, s, rb, lane, pair)
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x)
        split_pack_one(w, wf, o, rows, cols, KS, trans);
}

// Both forms a training step needs of one weight W (n_out, k) in ONE launch: the planes of W (forward, Y = X W^T) and of
// W^T packed from the same source (input gradient, dX = dY W).
__global__ void split_pack_both_kernel(const float* __restrict__ w, uint32_t* __restrict__ wf, uint32_t* __restrict__ wft, int n_out,
                                       int k, int KS, int chunks, int KSt, int chunks_t) {
    const int64_t total = (int64_t)chunks * KS * 4 * 64 * 4, total_t = (int64_t)chunks_t * KSt * 4 * 64 * 4;
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total + total_t; o += (int64_t)gridDim.x * blockDim.x) {
        if (o < total) split_pack_one(w, wf, o, n_out, k, KS, 0);
        else split_pack_one(w, wft, o - total, k, n_out, KSt, 1);
    }
}

struct LinSplitParams {
    const float* x;         // (T, K) row-major
    const uint4* wf;        // packed planes
    const float* bias;      // (N) or null
    float* y;               // (T, N) row-major
    int64_t tokens;
    int k, n_out, chunks, relu;
    int cpb;                    // output chunks per workgroup: blockIdx.y owns chunks [cpb * y, cpb * (y + 1))
    unsigned long long* dbg;    // GS_PROFILE builds: per-workgroup phase clocks
};
#ifdef GS_PROFILE
#define GS_MARK(i) do { if (tid == 0) p.dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define GS_MARK(i) do {} while (0)
#endif

template <int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void lin_split_kernel(const LinSplitParams p) {
    constexpr int G = KS % 4 == 0 ? 4 : 2;       // k-steps per LDS tile: 96 or 48 MFMAs between barriers
    constexpr int TILE = G * GS_KSTEP;           // uint4 per tile (48 KB / 24 KB)
    constexpr int NG = KS / G;                   // tiles per chunk
    constexpr int NLD = TILE / 256;              // LDS-DMA instructions per wave per tile
    __shared__ uint4 ring[3 * TILE];             // tile t lives in slot t % 3: one being read, one landed, one in flight
    __shared__ float bias_s[GS_MAX_N];           // an ordinary load in the loop would make hipcc drain the DMA (vmcnt(0))
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int64_t tok0 = ((int64_t)blockIdx.x * 4 + w) * 32;
    const int64_t tk = tok0 + j < p.tokens ? tok0 + j : p.tokens - 1;
    const uint4* src = p.wf + tid;
    const int ntiles = p.chunks * NG;
    // few tokens: the output chunks are spread over blockIdx.y as well (a 4096-token product is 32 workgroups otherwise)
    const int c0 = (int)blockIdx.y * p.cpb, c1 = min(p.chunks, c0 + p.cpb);
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)ring + w * 1024;
    auto issue = [&](int t, int slot) {          // tile t -> ring slot (the tile behind the last is the last again)
        const uint4* g = src + (size_t)(t < ntiles ? t : ntiles - 1) * TILE;
        const uint32_t l = ring_base + slot * (TILE * 16);
#pragma unroll
        for (int i = 0; i < NLD; ++i) glds16(g + 256 * i, l + 4096 * i);
    };
    static_assert(NLD <= G * 4, "one DMA instruction per MFMA group at most");
    GS_MARK(0);
    float x[KS * 8];
    {
        const float* row = p.x + tk * p.k + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(row + 16 * s), b = *reinterpret_cast<const float4*>(row + 16 * s + 4);
            x[8 * s] = a.x; x[8 * s + 1] = a.y; x[8 * s + 2] = a.z; x[8 * s + 3] = a.w;
            x[8 * s + 4] = b.x; x[8 * s + 5] = b.y; x[8 * s + 6] = b.z; x[8 * s + 7] = b.w;
        }
    }
    for (int i = tid; i < p.n_out; i += 256) bias_s[i] = p.bias ? p.bias[i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the row loads: nothing ordinary in flight beside the DMA below
    GS_MARK(1);
    issue(c0 * NG, 0);
    issue(c0 * NG + 1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    __builtin_amdgcn_s_barrier();
    GS_MARK(2);
    // rows of a 32x32 tile: token (i & 3) + 8 (i >> 2) + 4 h; column: output feature chunk * 128 + r * 32 + j.
    // Address = wave-uniform row base (scalar registers) + one per-lane offset: no vector address arithmetic per store.
    const uint32_t lane_off = (uint32_t)(4 * h * p.n_out + j);
    const int64_t left = p.tokens - tok0 - 4 * h;
    const bool whole = tok0 + 32 <= p.tokens;                 // wave-uniform
    // store number n (0..63) of a finished chunk: row block n / 16, accumulator register n % 16
    auto store_one = [&](const gs_f32x16 (&res)[4], int chunk, int n) {
        const int r = n >> 4, i = n & 15, t = (i & 3) + 8 * (i >> 2);
        float v = res[r][i] + bias_s[chunk * 128 + r * 32 + j];
        if (p.relu) v = fmaxf(v, 0.f);
        float* row = p.y + (tok0 + t) * p.n_out + chunk * 128 + r * 32;      // uniform
        if (whole) row[lane_off] = v;
        else if (t < left) row[lane_off] = v;
    };
    int ti = c0 * NG, slot = 0;                  // slot = (ti - c0 * NG) % 3
    gs_f32x16 done[4];                           // results of the previous chunk, stored under the next chunk's first tile
    for (int c = c0; c < c1; ++c) {
        gs_f32x16 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g, ++ti, slot = slot == 2 ? 0 : slot + 1) {
            // The 64 stores of the previous chunk are spread over the MFMA groups of this one: issued as one block,
            // the four waves (and every CU of the chip, in step) saturate the write path while the matrix pipe idles.
            if (c == 1 && g == 0) GS_MARK(4);
            // the DMA of tile ti + 2: one instruction behind each of the first NLD MFMA groups (a block of them in front
            // cost 57 cycles each with the matrix pipe idle)
            const uint4* dsrc = src + (size_t)(ti + 2 < ntiles ? ti + 2 : ntiles - 1) * TILE;
            const uint32_t ddst = ring_base + (slot == 0 ? 2 : slot - 1) * (TILE * 16);
            if (c == 1 && g == 0) GS_MARK(5);
            const uint4* buf = ring + slot * TILE + lane;
            // fragments of (k-step, row block) group q + 1 are read while the six MFMAs of group q run
            uint4 fa[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[0][pl] = buf[pl * 64];
#pragma unroll
            for (int sl = 0; sl < G; ++sl) {
                const int s = g * G + sl;
                if constexpr (KS > 16) {          // keep the f32 row: split again per chunk instead of 1.5x the registers
#pragma unroll
                    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(x[8 * s + q]));
                }
                uint32_t bh[4], bm[4], bl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) split2(x[8 * s + 2 * q], x[8 * s + 2 * q + 1], bh[q], bm[q], bl[q]);
                const gs_bf16x8 Xh = __builtin_bit_cast(gs_bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
                const gs_bf16x8 Xm = __builtin_bit_cast(gs_bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
                const gs_bf16x8 Xl = __builtin_bit_cast(gs_bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = sl * 4 + r;
                    if (q + 1 < G * 4) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) fa[(q + 1) & 1][pl] = buf[((q + 1) * 3 + pl) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const gs_bf16x8 Wh = __builtin_bit_cast(gs_bf16x8, fa[q & 1][0]);
                    const gs_bf16x8 Wm = __builtin_bit_cast(gs_bf16x8, fa[q & 1][1]);
                    const gs_bf16x8 Wl = __builtin_bit_cast(gs_bf16x8, fa[q & 1][2]);
                    // tokens are the A operand (rows of the tile), output features the B operand: a lane ends up with
                    // ONE output column of 16 token rows, so a store instruction writes 128 contiguous bytes per row.
                    // smallest terms first
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xl, Wh, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xh, Wl, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xm, Wm, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xm, Wh, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xh, Wm, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xh, Wh, acc[r], 0, 0, 0);
                    if (q < NLD) glds16(dsrc + 256 * q, ddst + 4096 * q);
                    if (c > c0) {
                        constexpr int GROUPS = KS * 4;                 // MFMA groups per chunk
                        const int gq = g * G * 4 + q;
#pragma unroll
                        for (int n = gq * 64 / GROUPS; n < (gq + 1) * 64 / GROUPS; ++n) store_one(done, c - 1, n);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // tile ti + 1 has landed (ti + 2 may still be in flight); everyone is done reading tile ti
            if (c == 1 && g == 0) GS_MARK(6);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            if (c == 1 && g == 0) GS_MARK(7);
            __builtin_amdgcn_s_barrier();
            if (c == 1 && g == 0) GS_MARK(8);
            if (c == 0 && g == NG - 1) GS_MARK(3);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) done[r] = acc[r];
    }
    GS_MARK(9);
#pragma unroll
    for (int n = 0; n < 64; ++n) store_one(done, c1 - 1, n);
    GS_MARK(10);
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_split_kernel: the same arithmetic as an LDS-tiled GEMM for ANY K (multiple of 16) and any N:
//   Y (T, N) [+]= X (T, K) W^T [+ bias] [relu].
// lin_split_kernel keeps a wave's rows in registers, which caps K at 512 and leaves one wave per SIMD (nothing overlaps
// its row loads and output stores).  Here a workgroup (4 waves, 2 x 2) owns a 128-token x 128-output tile and both
// operands stream through a 3-slot LDS ring in K = 16 steps by LDS-DMA: X as f32 rows of 64 bytes whose 16-byte chunks
// are XOR-swizzled through the SOURCE address (chunk c of row r sits in chunk slot c ^ ((r >> 2) & 3): the two
// ds_read_b128 of a fragment are conflict-free), W as the packed bf16 planes of split_pack_kernel.  A wave computes
// 64 x 64 (2 x 2 MFMA tiles): per K step it splits its two X fragments into bf16 terms (88 VALU) for 24 MFMAs.  60 KB of
// LDS and ~150 registers: two workgroups per CU, so one's stores and barrier waits sit under the other's MFMAs.
struct GemmSplitParams {
    const float* x;         // (T, K) row-major
    const uint4* wf;        // packed planes (split_pack_kernel)
    const float* bias;      // (N) or null
    float* y;               // (T, N) row-major
    int64_t tokens;
    int k, n_out, chunks, ksteps, relu, accum;
    // BatchNorm fused on either side (gemm_split_wide_kernel<PRO, EPI>, pda_gemm_split_bn):
    const float* in_mi;     // PRO: (2, K) mean | invstd of the INPUT channels; X is read as relu((x - mean) * invstd * gamma + beta)
    const float* in_g;      //      (K) gamma
    const float* in_b;      //      (K) beta
    double* partial;        // EPI 1: [token tile][2][n_out] per-column sum / sum of squares of this launch's output
    int pool_ns;            // EPI 2: y is (tokens / pool_ns, n_out), the max over each group of pool_ns consecutive token rows
    // PRO 2 (inference): X row of token t = relu(P[scene(t) * gn + idx[t]] + W_xyz (xyz[idx[t]] - centre(t)) + bias): the first layer
    // of a wide SA scale (sa_point_gather_kernel, csrc/sa_xyz_grad.hip) formed in the operand load.  x = P (b * gn, K) rows.
    const int32_t* g_idx;   // (tokens)
    const float* g_xyz;     // (b, gn, 3)
    const float* g_ctr;     // (tokens / g_ns, 3)
    const float* g_w;       // first-layer weight (K, g_ldw): columns 0..2 are the coordinates'
    const float* g_bias;    // (K)
    int gn, g_ns, g_m, g_ldw;
};

constexpr int GT_XCHUNKS = 128 * 4;                  // uint4 per X tile (128 rows x 64 bytes)
constexpr int GT_SLOT = GT_XCHUNKS + GS_KSTEP;       // + 768 uint4 of W planes = 20 KB per slot
constexpr int GW_TILE_U4 = (8 + 8) * 192;            // gemm_split_wide_kernel: uint4 per plane buffer (48 KB)

__global__ __launch_bounds__(256, 2)
void gemm_split_kernel(const GemmSplitParams p) {
    __shared__ uint4 ring[3 * GT_SLOT];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int wt = w & 1, wo = w >> 1;
    const int nc = (int)(blockIdx.x % (unsigned)p.chunks);          // output chunk fastest: the workgroups that share an
    const int64_t tok0 = (int64_t)(blockIdx.x / (unsigned)p.chunks) * 128;   // X tile run together (L2)
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)ring;
    // DMA roles.  X: instruction i = 2w + e covers LDS chunks 64 i .. 64 i + 63 (16 rows); W: instructions 3w .. 3w + 2.
    const float* xsrc[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int pch = 64 * (2 * w + e) + lane, row = pch >> 2, cl = pch & 3, c = cl ^ ((row >> 2) & 3);
        const int64_t tok = tok0 + row < p.tokens ? tok0 + row : p.tokens - 1;
        xsrc[e] = p.x + tok * p.k + 4 * c;
    }
    const uint4* wsrc = p.wf + (size_t)nc * p.ksteps * GS_KSTEP + 64 * 3 * w + lane;
    auto issue = [&](int s, int slot) {
        const int ss = s < p.ksteps ? s : p.ksteps - 1;      // behind the last step: the last tile again, never read
        const uint32_t base = ring_base + slot * (GT_SLOT * 16);
#pragma unroll
        for (int e = 0; e < 2; ++e) glds16(reinterpret_cast<const uint4*>(xsrc[e] + 16 * ss), base + (2 * w + e) * 1024);
#pragma unroll
        for (int e = 0; e < 3; ++e) glds16(wsrc + (size_t)ss * GS_KSTEP + 64 * e, base + GT_XCHUNKS * 16 + (3 * w + e) * 1024);
    };
    gs_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // fragment addresses inside a slot (uint4 units)
    int xoff[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int row = wt * 64 + tt * 32 + r;
#pragma unroll
        for (int e = 0; e < 2; ++e) xoff[tt][e] = row * 4 + ((2 * h + e) ^ ((row >> 2) & 3));
    }
    const int woff = GT_XCHUNKS + (wo * 2 * 3) * 64 + lane;
    issue(0, 0);
    issue(1, 1);
    int slot = 0;
    for (int s = 0; s < p.ksteps; ++s, slot = slot == 2 ? 0 : slot + 1) {
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // my part of tile s has landed (tile s + 1 may be in flight)
        __builtin_amdgcn_s_barrier();                          // everyone's has; everyone is done with tile s - 1
        asm volatile("" ::: "memory");
        issue(s + 2, slot == 0 ? 2 : slot - 1);
        const uint4* buf = ring + slot * GT_SLOT;
        uint4 wfr[2][3];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wfr[ot][pl] = buf[woff + (ot * 3 + pl) * 64];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const uint4 a = buf[xoff[tt][0]], b = buf[xoff[tt][1]];
            uint32_t bh[4], bm[4], bl[4];
#ifdef GS_NOSPLIT
            bh[0] = a.x; bh[1] = a.y; bh[2] = a.z; bh[3] = a.w; bm[0] = b.x; bm[1] = b.y; bm[2] = b.z; bm[3] = b.w;
            bl[0] = a.x ^ b.x; bl[1] = a.y ^ b.y; bl[2] = a.z ^ b.z; bl[3] = a.w ^ b.w;
#else
            split2(__uint_as_float(a.x), __uint_as_float(a.y), bh[0], bm[0], bl[0]);
            split2(__uint_as_float(a.z), __uint_as_float(a.w), bh[1], bm[1], bl[1]);
            split2(__uint_as_float(b.x), __uint_as_float(b.y), bh[2], bm[2], bl[2]);
            split2(__uint_as_float(b.z), __uint_as_float(b.w), bh[3], bm[3], bl[3]);
#endif
            const gs_bf16x8 Xh = __builtin_bit_cast(gs_bf16x8, make_uint4(bh[0], bh[1], bh[2], bh[3]));
            const gs_bf16x8 Xm = __builtin_bit_cast(gs_bf16x8, make_uint4(bm[0], bm[1], bm[2], bm[3]));
            const gs_bf16x8 Xl = __builtin_bit_cast(gs_bf16x8, make_uint4(bl[0], bl[1], bl[2], bl[3]));
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                const gs_bf16x8 Wh = __builtin_bit_cast(gs_bf16x8, wfr[ot][0]);
                const gs_bf16x8 Wm = __builtin_bit_cast(gs_bf16x8, wfr[ot][1]);
                const gs_bf16x8 Wl = __builtin_bit_cast(gs_bf16x8, wfr[ot][2]);
                gs_f32x16 c = acc[tt][ot];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xl, Wh, c, 0, 0, 0);    // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xh, Wl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xm, Wm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xm, Wh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xh, Wm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Xh, Wh, c, 0, 0, 0);
                acc[tt][ot] = c;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the two dummy tiles: no DMA may outlive the wave's LDS
    // rows of a 32 x 32 tile: token (i & 3) + 8 (i >> 2) + 4 h; column: output nc * 128 + (2 wo + ot) * 32 + r
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        const int col = nc * 128 + (wo * 2 + ot) * 32 + r;
        if (col >= p.n_out) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int64_t trow = tok0 + wt * 64 + tt * 32 + 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t tok = trow + (i & 3) + 8 * (i >> 2);
                if (tok < p.tokens) {
                    float* dst = p.y + tok * p.n_out + col;
                    float v = acc[tt][ot][i] + bias;
                    if (p.accum) v += *dst;
                    if (p.relu) v = fmaxf(v, 0.f);
#ifdef GS_NOSTORE
                    if (v == 12345.678f)
#endif
                    *dst = v;
                }
            }
        }
    }
}

// gemm_split_wide_kernel: 8 waves (4 x 2) per workgroup = 256 tokens x 256 outputs, a wave computes 64 x 128 (2 x 4 MFMA
// tiles, 128 accumulator registers), with the loop structure of wgrad_split_kernel (csrc/wgrad.hip):
//   * producer role: wave w owns tokens 32 w .. 32 w + 31 of the tile: lane (r, h) loads X[token r][16 s + 8 h .. + 7]
//     (two 16-byte loads, a step ahead in registers), splits them -- every X value ONCE per workgroup instead of once
//     per wave that uses it -- and stores the three planes as ready MFMA fragments; the packed W planes of the step
//     (24 KB, fragment-ordered by split_pack_kernel) go global -> registers -> LDS, three 16-byte pieces per lane;
//   * consumer role: wave (wt, wo) computes 64 tokens x 128 outputs: 18 ds_read_b128 for 48 MFMAs per K step.
// Two plane buffers of 48 KB, one barrier per K step; the last quarter of a step's MFMAs is issued behind the NEXT step's
// barrier (over the latency of its first fragment reads), the split in four pieces behind three MFMAs each, stores and
// loads one per three MFMAs.  K a multiple of 32: the steps come in pairs (the register sets swap roles).
// Measured against the form this replaces (both operands through a 3-slot LDS-DMA ring, X split by every wave that reads
// it: bf16 pipe 52 % busy; an LDS-DMA piece costs 100-185 issue cycles inside a busy phase, MI355X_MICROARCH.md): 3-9 %
// less time on the step's shapes.  Four waves per workgroup and two workgroups per CU (128 x 256 tiles) measured 3-8 %
// slower than this form: the output stores are not what holds it back.
typedef uint32_t gp_u32x4 __attribute__((ext_vector_type(4)));

// PRO = 1: the X operand is the PRE-BatchNorm tensor of the layer before; the producer role applies that layer's training-mode
//   BatchNorm + ReLU (the expression of bn_apply_kernel, csrc/bn_relu.hip, bit for bit) to the eight values it is about to
//   split, so relu(bn(x)) never exists in HBM.  The four per-channel constants sit in LDS behind the plane buffers.
// EPI = 1: besides storing Y, the workgroup writes the per-column sum and sum of squares of its 256 token rows (double,
//   fixed order: 32 rows per lane, the two half-waves, then the four token-waves) -- the statistics pass of the BatchNorm
//   that follows, without reading Y again.  EPI = 2 (inference): only the max over every group of pool_ns consecutive token
//   rows leaves the kernel (the max-pool over nsample behind the last layer of an SA scale).  (The reduction pass of the BatchNorm BACKWARD in the epilogue of the
//   input-gradient GEMM, z read tile by tile next to the stores, was measured and removed: 131072 x 512 -> 512 went from
//   0.36 to 0.60 ms, the standalone pass costs 0.10.)
template <int PRO, int EPI>
__global__ __launch_bounds__(512, 1)
void gemm_split_wide_kernel(const GemmSplitParams p) {
    constexpr int TW = 4;                           // token-waves (x 2 output-waves)
    constexpr int XF = 2 * TW;                      // X fragments (32 tokens each) per tile
    constexpr int NP = 24 / (2 * TW);               // W pieces (1 KB) per wave and step
    constexpr int TILE_U4 = GW_TILE_U4;             // uint4 per plane buffer: X fragments, then W pieces 0..23
    extern __shared__ uint4 gp_planes[];            // 2 * TILE_U4
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int wt = w % TW, wo = w / TW;
    const int wide_chunks = (p.chunks + 1) >> 1;
    const int ncb = (int)(blockIdx.x % (unsigned)wide_chunks);
    const int64_t tok0 = (int64_t)(blockIdx.x / (unsigned)wide_chunks) * (64 * TW);
    // producer sources
    const int64_t ptok = tok0 + 32 * w + r < p.tokens ? tok0 + 32 * w + r : p.tokens - 1;
    const float* xsrc = p.x + ptok * p.k + 8 * h;
    float gdx = 0.f, gdy = 0.f, gdz = 0.f;                            // PRO 2: this lane's token relative to its centre
    if constexpr (PRO == 2) {
        const int64_t grp = ptok / p.g_ns;
        const int64_t prow = (grp / p.g_m) * p.gn + p.g_idx[ptok];
        xsrc = p.x + prow * p.k + 8 * h;
        const float* pt = p.g_xyz + prow * 3;
        const float* ct = p.g_ctr + grp * 3;
        gdx = pt[0] - ct[0]; gdy = pt[1] - ct[1]; gdz = pt[2] - ct[2];      // pointnet2_utils.py:692
    }
    const uint4* wsrc[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) {
        const int piece = NP * w + e, half = piece / 12, j = piece - 12 * half;
        int chunk = 2 * ncb + half;
        if (chunk >= p.chunks) chunk = p.chunks - 1;      // an odd chunk count repeats the last chunk (never stored)
        wsrc[e] = p.wf + (size_t)chunk * p.ksteps * GS_KSTEP + 64 * j + lane;
    }
    auto wld = [&](int e, int ss) { return *reinterpret_cast<const gp_u32x4*>(wsrc[e] + (size_t)ss * GS_KSTEP); };
    auto wst = [&](uint4* dst, const gp_u32x4& v) { *reinterpret_cast<gp_u32x4*>(dst) = v; };
    auto load = [&](int st, float4 (&x)[2], gp_u32x4 (&wv)[NP]) {
        const int ss = st < p.ksteps ? st : p.ksteps - 1;              // behind the last step: the last step again, never used
        x[0] = *reinterpret_cast<const float4*>(xsrc + 16 * ss);
        x[1] = *reinterpret_cast<const float4*>(xsrc + 16 * ss + 4);
#pragma unroll
        for (int e = 0; e < NP; ++e) wv[e] = wld(e, ss);
    };
    uint4* const my_x = gp_planes + w * 192 + lane;                     // + buffer * TILE_U4 + plane * 64
    uint4* const my_w = gp_planes + XF * 192 + (NP * w) * 64 + lane;    // + buffer * TILE_U4 + e * 64
    // PRO: {mean, invstd, gamma, beta} of input channel k at cst[k]
    float4* const cst = reinterpret_cast<float4*>(gp_planes + 2 * TILE_U4);
    if constexpr (PRO == 1) {
        for (int k = tid; k < p.k; k += 512) cst[k] = make_float4(p.in_mi[k], p.in_mi[p.k + k], p.in_g[k], p.in_b[k]);
        __syncthreads();
    }
    if constexpr (PRO == 2) {
        for (int k = tid; k < p.k; k += 512) {
            const float* wr = p.g_w + (size_t)k * p.g_ldw;
            cst[k] = make_float4(wr[0], wr[1], wr[2], p.g_bias ? p.g_bias[k] : 0.f);
        }
        __syncthreads();
    }
    auto bnrelu = [&](float x, const float4& c) {
        if constexpr (PRO == 2)      // sa_point_gather_kernel's expression, bit for bit
            return fmaxf(__builtin_fmaf(c.z, gdz, __builtin_fmaf(c.y, gdy, __builtin_fmaf(c.x, gdx, x))) + c.w, 0.f);
        else
            return fmaxf((x - c.x) * c.y * c.z + c.w, 0.f);
    };
    const float4* const my_cst = cst + 8 * h;                           // + 16 * step + j
    gs_f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int aoff = (2 * wt) * 192 + lane, boff = (XF + 4 * wo) * 192 + lane;
    gs_bf16x8 Bf[2][3];
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // the six products, smallest terms first

    // One step (issue order = source order, sched_barrier fences).  xa / wa: the raw values of step st + 1 (split and
    // stored here), xb / wb receive those of step st + 2.  Ap: the previous step's token fragments, An: this step's.
    auto step = [&](int st, const float4 (&xa)[2], const gp_u32x4 (&wa)[NP], float4 (&xb)[2], gp_u32x4 (&wb)[NP],
                    const gs_bf16x8 (&Ap)[2][3], gs_bf16x8 (&An)[2][3]) {
        __syncthreads();                   // planes of step st are complete; everyone is done with step st - 1
        const uint4* buf = gp_planes + (st & 1) * TILE_U4;
        const int ob = ((st + 1) & 1) * TILE_U4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) An[i][pl] = __builtin_bit_cast(gs_bf16x8, buf[aoff + i * 192 + pl * 64]);
        auto read_b = [&](int jt) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) Bf[jt & 1][pl] = __builtin_bit_cast(gs_bf16x8, buf[boff + jt * 192 + pl * 64]);
        };
        auto mfma3 = [&](const gs_bf16x8 (&A)[2][3], int jt, int k) {     // MFMAs 3k .. 3k + 2 of the 12 of output block jt
#pragma unroll
            for (int e = 3 * k; e < 3 * k + 3; ++e) {
                const int i = e & 1, pr = e >> 1;
                acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][PA[pr]], Bf[jt & 1][PB[pr]], acc[i][jt], 0, 0, 0);
            }
        };
        read_b(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) mfma3(Ap, 3, q);                       // the previous step's output block 3
        __builtin_amdgcn_sched_barrier(0);
        read_b(1);
        uint32_t ph[4], pm[4], pl[4];
        const float xv[8] = {xa[0].x, xa[0].y, xa[0].z, xa[0].w, xa[1].x, xa[1].y, xa[1].z, xa[1].w};
        const int kn = st + 1 < p.ksteps ? st + 1 : p.ksteps - 1;        // the step whose planes are made here
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 c0, c1;
            if constexpr (PRO) { c0 = my_cst[16 * kn + 2 * q]; c1 = my_cst[16 * kn + 2 * q + 1]; }
            mfma3(An, 0, q);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PRO) split2(bnrelu(xv[2 * q], c0), bnrelu(xv[2 * q + 1], c1), ph[q], pm[q], pl[q]);
            else split2(xv[2 * q], xv[2 * q + 1], ph[q], pm[q], pl[q]);
            __builtin_amdgcn_sched_barrier(0);
        }
        my_x[ob] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        my_x[ob + 64] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        my_x[ob + 128] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        read_b(2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                      // the W pieces of step st + 1: registers -> LDS
            mfma3(An, 1, q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < NP; ++e)
                if (e * 4 / NP == q) wst(my_w + ob + 64 * e, wa[e]);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_b(3);                          // stays in Bf[1] for the MFMAs behind the next barrier
        const int ss = st + 2 < p.ksteps ? st + 2 : p.ksteps - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                      // the loads of step st + 2
            mfma3(An, 2, q);
            __builtin_amdgcn_sched_barrier(0);
            if (q == 0) {
                xb[0] = *reinterpret_cast<const float4*>(xsrc + 16 * ss);
                xb[1] = *reinterpret_cast<const float4*>(xsrc + 16 * ss + 4);
            }
#pragma unroll
            for (int e = 0; e < NP; ++e)
                if ((e + 1) * 4 / (NP + 1) == q) wb[e] = wld(e, ss);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    float4 x1[2], x2[2];
    gp_u32x4 w1[NP], w2[NP];
    gs_bf16x8 A0[2][3], A1[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            A0[i][pl] = __builtin_bit_cast(gs_bf16x8, make_uint4(0u, 0u, 0u, 0u));      // "the step before the first": zeros
            Bf[1][pl] = A0[i][pl];
        }
    load(0, x1, w1);
    {                                                                   // the first step's planes
        uint32_t ph[4], pm[4], pl[4];
        const float xv[8] = {x1[0].x, x1[0].y, x1[0].z, x1[0].w, x1[1].x, x1[1].y, x1[1].z, x1[1].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (PRO) split2(bnrelu(xv[2 * q], my_cst[2 * q]), bnrelu(xv[2 * q + 1], my_cst[2 * q + 1]), ph[q], pm[q], pl[q]);
            else split2(xv[2 * q], xv[2 * q + 1], ph[q], pm[q], pl[q]);
        }
        my_x[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        my_x[64] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        my_x[128] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
#pragma unroll
        for (int e = 0; e < NP; ++e) wst(my_w + 64 * e, w1[e]);
    }
    load(1, x1, w1);
    for (int st = 0; st < p.ksteps; st += 2) {
        step(st, x1, w1, x2, w2, A0, A1);
        step(st + 1, x2, w2, x1, w1, A1, A0);
    }
#pragma unroll
    for (int e = 0; e < 12; ++e)                   // output block 3 of the last step
        acc[e & 1][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0[e & 1][PA[e >> 1]], Bf[1][PB[e >> 1]], acc[e & 1][3], 0, 0, 0);
    // rows of a 32 x 32 tile: token (i & 3) + 8 (i >> 2) + 4 h; column: output (2 ncb + wo) * 128 + ot * 32 + r
    if constexpr (EPI == 2) {
        // Pooled output: out[group][col] = max over the group's p.pool_ns consecutive token rows of relu?(acc + bias); Y itself is
        // never written.  pool_ns in {16, 32, 64} divides the 64 tokens of a wave, so a group lives in one wave: eight / sixteen /
        // thirty-two registers of a lane (rows (i & 3) + 8 (i >> 2) + 4 h cover 16 consecutive rows for i >> 3 fixed), then the
        // other half-wave.
        const int ns = p.pool_ns;
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const int col = (2 * ncb + wo) * 128 + ot * 32 + r;
            const float bias = (p.bias && col < p.n_out) ? p.bias[col] : 0.f;
            float g16[4];                           // the four groups of 16 rows of this wave's 64 tokens
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float m = -__builtin_inff();
#pragma unroll
                    for (int i = 8 * half; i < 8 * half + 8; ++i) {
                        float v = acc[tt][ot][i] + bias;
                        if (p.relu) v = fmaxf(v, 0.f);
                        m = fmaxf(m, v);
                    }
                    g16[2 * tt + half] = fmaxf(m, __shfl_xor(m, 32));
                }
            if (h == 0 && col < p.n_out) {
                const int64_t t0 = tok0 + wt * 64;                  // first token of the wave
                if (ns == 16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (t0 + 16 * q < p.tokens) p.y[((t0 + 16 * q) >> 4) * p.n_out + col] = g16[q];
                } else if (ns == 32) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (t0 + 32 * q < p.tokens) p.y[((t0 + 32 * q) >> 5) * p.n_out + col] = fmaxf(g16[2 * q], g16[2 * q + 1]);
                } else if (t0 < p.tokens) {
                    p.y[(t0 >> 6) * p.n_out + col] = fmaxf(fmaxf(g16[0], g16[1]), fmaxf(g16[2], g16[3]));
                }
            }
        }
        return;
    }
    double cs1[4], cs2[4];                          // EPI: this lane's column sums (one column per output block)
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
        cs1[ot] = 0; cs2[ot] = 0;
        const int col = (2 * ncb + wo) * 128 + ot * 32 + r;
        if (col >= p.n_out) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int64_t trow = tok0 + wt * 64 + tt * 32 + 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t tok = trow + (i & 3) + 8 * (i >> 2);
                if (tok < p.tokens) {
                    float* dst = p.y + tok * p.n_out + col;
                    float v = acc[tt][ot][i] + bias;
                    if (p.accum) v += *dst;
                    if (p.relu) v = fmaxf(v, 0.f);
#ifdef GS_NOSTORE
                    if (v == 12345.678f)
#endif
                    *dst = v;
                    if constexpr (EPI == 1) { cs1[ot] += v; cs2[ot] += (double)v * v; }
                }
            }
        }
    }
    if constexpr (EPI == 1) {
        // lanes r and r + 32 hold the same columns; the four token-waves of an output-wave too: LDS [wt][column 0..255][2]
        double* red = reinterpret_cast<double*>(gp_planes);
        __syncthreads();                            // every wave is done with the plane buffers
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const double a = cs1[ot] + __shfl_xor(cs1[ot], 32), b = cs2[ot] + __shfl_xor(cs2[ot], 32);
            if (h == 0) {
                const int cl = wo * 128 + ot * 32 + r;
                red[(wt * 256 + cl) * 2] = a; red[(wt * 256 + cl) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < 256) {
            const int col = 2 * ncb * 128 + tid;
            if (col < p.n_out) {
                double a = 0, b = 0;
#pragma unroll
                for (int q = 0; q < TW; ++q) { a += red[(q * 256 + tid) * 2]; b += red[(q * 256 + tid) * 2 + 1]; }
                double* pp = p.partial + (size_t)(blockIdx.x / (unsigned)wide_chunks) * 2 * p.n_out;
                pp[col] = a; pp[p.n_out + col] = b;
            }
        }
    }
}

}  // namespace pda

PDA_API int64_t pda_linear_split_packed_bytes(int n_out, int k) {
    if (n_out <= 0 || k <= 0) return 0;
    return (int64_t)pda::divup(n_out, 128) * pda::divup(k, 32) * 2 * 12 * 64 * 16;
}

PDA_API int pda_linear_split_pack(const float* w, void* wf, int n_out, int k, int transposed_source, pda_stream_t stream) {
    PDA_REQUIRE(w && wf && n_out > 0 && k > 0, "pda_linear_split_pack: bad argument");
    const int chunks = pda::divup(n_out, 128), KS = pda::divup(k, 32) * 2;
    const int64_t total = (int64_t)chunks * KS * 4 * 64 * 4;
    const int blocks = (int)(pda::divup64(total, 256) < 2048 ? pda::divup64(total, 256) : 2048);
    hipLaunchKernelGGL(pda::split_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (uint32_t*)wf, n_out, k, KS,
                       chunks, transposed_source);
    return pda::check_launch("pda_linear_split_pack");
}

PDA_API int pda_linear_split_pack_both(const float* w, void* wf, void* wft, int n_out, int k, pda_stream_t stream) {
    PDA_REQUIRE(w && wf && wft && n_out > 0 && k > 0, "pda_linear_split_pack_both: bad argument");
    const int chunks = pda::divup(n_out, 128), KS = pda::divup(k, 32) * 2;
    const int chunks_t = pda::divup(k, 128), KSt = pda::divup(n_out, 32) * 2;
    const int64_t total = ((int64_t)chunks * KS + (int64_t)chunks_t * KSt) * 4 * 64 * 4;
    const int blocks = (int)(pda::divup64(total, 256) < 2048 ? pda::divup64(total, 256) : 2048);
    hipLaunchKernelGGL(pda::split_pack_both_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (uint32_t*)wf, (uint32_t*)wft,
                       n_out, k, KS, chunks, KSt, chunks_t);
    return pda::check_launch("pda_linear_split_pack_both");
}

PDA_API int pda_linear_split(const float* x, const void* wf, const float* bias, float* y, int64_t tokens, int k, int n_out,
                             int relu, pda_stream_t stream) {
    PDA_REQUIRE(tokens >= 0 && k > 0 && n_out > 0, "pda_linear_split: bad size");
    if (tokens == 0) return PDA_OK;
    PDA_REQUIRE(x && wf && y && (((uintptr_t)x | (uintptr_t)wf | (uintptr_t)y | (uintptr_t)bias) & 15) == 0,
                "pda_linear_split: null or misaligned pointer");
    if (k % 32 != 0 || k > 512 || n_out % 128 != 0 || n_out > pda::GS_MAX_N) {
        pda::set_error("pda_linear_split: no kernel built for K=%d, N=%d (K a multiple of 32 <= 512, N a multiple of 128 <= 2048)", k, n_out);
        return PDA_ERR_UNSUPPORTED;
    }
    pda::LinSplitParams p{};
    p.x = x; p.wf = (const uint4*)wf; p.bias = bias; p.y = y; p.tokens = tokens; p.k = k; p.n_out = n_out; p.chunks = n_out / 128;
    p.relu = relu;
#ifdef GS_PROFILE
    p.dbg = (unsigned long long*)bias; p.bias = nullptr;
#endif
    // one workgroup walks all chunks of its 128 tokens (X read once) unless that leaves most of the chip idle
    const int64_t tiles = pda::divup64(tokens, 128);
    const int want = tiles >= 192 ? 1 : (int)pda::divup64(256, tiles);
    p.cpb = pda::divup(p.chunks, want < p.chunks ? want : p.chunks);
    const dim3 grid((unsigned)tiles, (unsigned)pda::divup(p.chunks, p.cpb)), block(256);
    const hipStream_t s = (hipStream_t)stream;
    switch (k / 16) {
#define PDA_GS_CASE(KS) case KS: hipLaunchKernelGGL((pda::lin_split_kernel<KS>), grid, block, 0, s, p); break
        PDA_GS_CASE(2); PDA_GS_CASE(4); PDA_GS_CASE(6); PDA_GS_CASE(8); PDA_GS_CASE(12); PDA_GS_CASE(16); PDA_GS_CASE(24); PDA_GS_CASE(32);
#undef PDA_GS_CASE
        default:
            pda::set_error("pda_linear_split: no kernel built for K=%d", k);
            return PDA_ERR_UNSUPPORTED;
    }
    return pda::check_launch("pda_linear_split");
}

PDA_API int pda_gemm_split(const float* x, const void* wf, const float* bias, float* y, int64_t tokens, int k, int n_out,
                           int relu, int accumulate, pda_stream_t stream) {
    PDA_REQUIRE(tokens >= 0 && k > 0 && n_out > 0, "pda_gemm_split: bad size");
    if (tokens == 0) return PDA_OK;
    PDA_REQUIRE(x && wf && y && (((uintptr_t)x | (uintptr_t)wf) & 15) == 0, "pda_gemm_split: null or misaligned pointer");
    if (k % 16 != 0) {
        pda::set_error("pda_gemm_split: K=%d must be a multiple of 16", k);
        return PDA_ERR_UNSUPPORTED;
    }
    pda::GemmSplitParams p{};
    p.x = x; p.wf = (const uint4*)wf; p.bias = bias; p.y = y; p.tokens = tokens; p.k = k; p.n_out = n_out;
    p.chunks = pda::divup(n_out, 128); p.ksteps = pda::divup(k, 32) * 2; p.relu = relu; p.accum = accumulate;
    // the packed planes hold ceil(K / 32) * 2 K steps (zero beyond K); X is only read up to K: walk K / 16 steps
    const int64_t blocks = pda::divup64(tokens, 128) * p.chunks;
    PDA_REQUIRE(blocks < (1ll << 31), "pda_gemm_split: too many tiles");
    const int packed_steps = p.ksteps;
    p.ksteps = k / 16;
    // the W tiles of a chunk are packed_steps apart
    if (packed_steps != p.ksteps) {
        pda::set_error("pda_gemm_split: K=%d must be a multiple of 32 (packed planes come in pairs of K steps)", k);
        return PDA_ERR_UNSUPPORTED;
    }
    // 256 x 256 tiles once they fill the chip; PDA_GEMM_SPLIT_TILE=128|256 forces a form
    static const int force = getenv("PDA_GEMM_SPLIT_TILE") ? atoi(getenv("PDA_GEMM_SPLIT_TILE")) : 0;
    const int64_t wide_blocks = pda::divup64(tokens, 256) * ((p.chunks + 1) / 2);
    if (force == 256 || (force != 128 && wide_blocks >= 200 && (p.chunks % 2 == 0 || p.chunks >= 5))) {
        static pda::PerDevice<bool> lds_ok;
        const bool ok = lds_ok.get([] { return hipFuncSetAttribute((const void*)pda::gemm_split_wide_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                   2 * pda::GW_TILE_U4 * 16) == hipSuccess; });
        if (ok) {
            hipLaunchKernelGGL((pda::gemm_split_wide_kernel<0, 0>), dim3((unsigned)wide_blocks), dim3(512), 2 * pda::GW_TILE_U4 * 16,
                               (hipStream_t)stream, p);
            return pda::check_launch("pda_gemm_split");
        }
    }
    hipLaunchKernelGGL(pda::gemm_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return pda::check_launch("pda_gemm_split");
}

// Y = relu(bn_in(X)) W^T (in_* given) or X W^T, with the statistics pass of the BatchNorm behind it in the epilogue
// (stats_mode 1).  Always the 256 x 256 tile kernel: partial holds pda_gemm_split_bn_tiles(tokens) x 2 x n_out doubles.
PDA_API int64_t pda_gemm_split_bn_tiles(int64_t tokens) { return tokens > 0 ? pda::divup64(tokens, 256) : 0; }

PDA_API int pda_gemm_split_bn(const float* x, const void* wf, float* y, int64_t tokens, int k, int n_out, const float* in_mean_invstd,
                              const float* in_gamma, const float* in_beta, int stats_mode, double* partial, pda_stream_t stream) {
    PDA_REQUIRE(tokens >= 1 && k > 0 && n_out > 0, "pda_gemm_split_bn: bad size");
    PDA_REQUIRE(x && wf && y && (((uintptr_t)x | (uintptr_t)wf) & 15) == 0, "pda_gemm_split_bn: null or misaligned pointer");
    const bool pro = in_mean_invstd != nullptr;
    PDA_REQUIRE(pro == (in_gamma != nullptr) && pro == (in_beta != nullptr), "pda_gemm_split_bn: the input BatchNorm needs mean_invstd, gamma and beta");
    PDA_REQUIRE(stats_mode >= 0 && stats_mode <= 1 && (stats_mode == 0 || partial), "pda_gemm_split_bn: stats_mode=%d (0 or 1; 1 needs `partial`)", stats_mode);
    if (k % 32 != 0 || k > 1024) {
        pda::set_error("pda_gemm_split_bn: no kernel built for K=%d (a multiple of 32, <= 1024)", k);
        return PDA_ERR_UNSUPPORTED;
    }
    pda::GemmSplitParams p{};
    p.x = x; p.wf = (const uint4*)wf; p.y = y; p.tokens = tokens; p.k = k; p.n_out = n_out;
    p.chunks = pda::divup(n_out, 128); p.ksteps = k / 16;
    p.in_mi = in_mean_invstd; p.in_g = in_gamma; p.in_b = in_beta; p.partial = partial;
    const int64_t wide_blocks = pda::divup64(tokens, 256) * ((p.chunks + 1) / 2);
    PDA_REQUIRE(wide_blocks < (1ll << 31), "pda_gemm_split_bn: too many tiles");
    const int lds = 2 * pda::GW_TILE_U4 * 16 + (pro ? k * 16 : 0);
    const void* fn;
    if (pro && stats_mode == 1) fn = (const void*)pda::gemm_split_wide_kernel<1, 1>;
    else if (pro && stats_mode == 0) fn = (const void*)pda::gemm_split_wide_kernel<1, 0>;
    else if (stats_mode == 1) fn = (const void*)pda::gemm_split_wide_kernel<0, 1>;
    else fn = (const void*)pda::gemm_split_wide_kernel<0, 0>;
    static pda::PerDevice<bool> lds_ok[4];
    const bool ok = lds_ok[(pro ? 2 : 0) + stats_mode].get([fn] {
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * pda::GW_TILE_U4 * 16 + 1024 * 16) == hipSuccess; });
    PDA_REQUIRE(ok, "pda_gemm_split_bn: dynamic LDS refused");
    void* args[] = {(void*)&p};
    PDA_REQUIRE(hipLaunchKernel(fn, dim3((unsigned)wide_blocks), dim3(512), args, lds, (hipStream_t)stream) == hipSuccess,
                "pda_gemm_split_bn: launch failed");
    return pda::check_launch("pda_gemm_split_bn");
}

// out (tokens / ns, n_out) = max over each group of ns consecutive token rows of relu?(x W^T + bias): the last layer of an SA
// scale in inference with the max over nsample in the epilogue (ns in {16, 32, 64}; tokens a multiple of ns).
PDA_API int pda_gemm_split_maxpool(const float* x, const void* wf, const float* bias, float* out, int64_t tokens, int k, int n_out, int ns,
                                   int relu, pda_stream_t stream) {
    PDA_REQUIRE(tokens >= 0 && k > 0 && n_out > 0, "pda_gemm_split_maxpool: bad size");
    if (tokens == 0) return PDA_OK;
    PDA_REQUIRE(x && wf && out && (((uintptr_t)x | (uintptr_t)wf) & 15) == 0, "pda_gemm_split_maxpool: null or misaligned pointer");
    if (k % 32 != 0 || (ns != 16 && ns != 32 && ns != 64) || tokens % ns != 0) {
        pda::set_error("pda_gemm_split_maxpool: K=%d (a multiple of 32), ns=%d (16, 32 or 64), tokens=%lld (a multiple of ns)", k, ns, (long long)tokens);
        return PDA_ERR_UNSUPPORTED;
    }
    pda::GemmSplitParams p{};
    p.x = x; p.wf = (const uint4*)wf; p.bias = bias; p.y = out; p.tokens = tokens; p.k = k; p.n_out = n_out;
    p.chunks = pda::divup(n_out, 128); p.ksteps = k / 16; p.relu = relu; p.pool_ns = ns;
    const int64_t wide_blocks = pda::divup64(tokens, 256) * ((p.chunks + 1) / 2);
    PDA_REQUIRE(wide_blocks < (1ll << 31), "pda_gemm_split_maxpool: too many tiles");
    static pda::PerDevice<bool> lds_ok;
    const bool ok = lds_ok.get([] { return hipFuncSetAttribute((const void*)pda::gemm_split_wide_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                               2 * pda::GW_TILE_U4 * 16) == hipSuccess; });
    PDA_REQUIRE(ok, "pda_gemm_split_maxpool: dynamic LDS refused");
    hipLaunchKernelGGL((pda::gemm_split_wide_kernel<0, 2>), dim3((unsigned)wide_blocks), dim3(512), 2 * pda::GW_TILE_U4 * 16, (hipStream_t)stream, p);
    return pda::check_launch("pda_gemm_split_maxpool");
}

// Inference, the first two layers of a wide SA scale in one launch: y (tokens, n_out) = relu?(A W2^T + bias2) with
// A[token] = relu(P[idx[token]] + W1_xyz (xyz[idx[token]] - centre) + bias1) formed in the operand load (what pda_sa_point_gather
// would write).  point_rows (b * n, K) = the per-point projection of the features; w1 (K, ldw1) the first layer's weight.
PDA_API int pda_gemm_split_gather(const float* point_rows, const float* xyz, const float* new_xyz, const int32_t* idx, const float* w1, int ldw1,
                                  const float* bias1, const void* wf, const float* bias2, float* y, int b, int n, int m, int ns, int k, int n_out,
                                  int relu, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 1 && m >= 0 && ns >= 1 && k > 0 && n_out > 0 && ldw1 >= 3, "pda_gemm_split_gather: bad size");
    const int64_t tokens = (int64_t)b * m * ns;
    if (tokens == 0) return PDA_OK;
    PDA_REQUIRE(point_rows && xyz && new_xyz && idx && w1 && wf && y && (((uintptr_t)point_rows | (uintptr_t)wf) & 15) == 0,
                "pda_gemm_split_gather: null or misaligned pointer");
    if (k % 32 != 0 || k > 1024) {
        pda::set_error("pda_gemm_split_gather: no kernel built for K=%d (a multiple of 32, <= 1024)", k);
        return PDA_ERR_UNSUPPORTED;
    }
    pda::GemmSplitParams p{};
    p.x = point_rows; p.wf = (const uint4*)wf; p.bias = bias2; p.y = y; p.tokens = tokens; p.k = k; p.n_out = n_out;
    p.chunks = pda::divup(n_out, 128); p.ksteps = k / 16; p.relu = relu;
    p.g_idx = idx; p.g_xyz = xyz; p.g_ctr = new_xyz; p.g_w = w1; p.g_bias = bias1; p.gn = n; p.g_ns = ns; p.g_m = m; p.g_ldw = ldw1;
    const int64_t wide_blocks = pda::divup64(tokens, 256) * ((p.chunks + 1) / 2);
    PDA_REQUIRE(wide_blocks < (1ll << 31), "pda_gemm_split_gather: too many tiles");
    static pda::PerDevice<bool> lds_ok;
    const bool ok = lds_ok.get([] { return hipFuncSetAttribute((const void*)pda::gemm_split_wide_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                               2 * pda::GW_TILE_U4 * 16 + 1024 * 16) == hipSuccess; });
    PDA_REQUIRE(ok, "pda_gemm_split_gather: dynamic LDS refused");
    hipLaunchKernelGGL((pda::gemm_split_wide_kernel<2, 0>), dim3((unsigned)wide_blocks), dim3(512), 2 * pda::GW_TILE_U4 * 16 + k * 16,
                       (hipStream_t)stream, p);
    return pda::check_launch("pda_gemm_split_gather");
}
