// sa_mlp.hip -- fused "group -> shared MLP -> max-pool" of one scale of a vanilla
// set-abstraction layer (inference form: BatchNorm folded into a per-channel scale/shift).
//
// Replaces, for one scale, the reference's chain of separate kernels
//   QueryAndGroup (pointnet2_utils.py:671-704: 2x group_points + centre subtraction + cat)
//   -> [Conv2d 1x1 (no bias) -> BatchNorm2d -> ReLU] x 3 -> F.max_pool2d over nsample
//   (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:1657-1670),
// which materialises every (B, C, npoint, nsample) intermediate in HBM (134 MB per tensor at
// ONCE layer 5).  Here only idx, the gathered inputs and the pooled (B, C_out, npoint) move.
//
// Structure (MI355X: 4 waves per workgroup, ONE wave per SIMD owning the whole 512-register
// file):
//   * a wave owns 32 columns = 32 consecutive (centre, sample) slots.  It gathers the grouped
//     input [dx,dy,dz | features] of its columns straight into registers in MFMA B-operand
//     layout (lane half h holds input channels k = 2t + h for k-step t);
//   * every layer is D = W x H on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-exact a
//     k-ordered fmaf chain, same precision as the reference's fp32 conv; MI355X peak 157 TF);
//   * the 32x32 accumulator of one layer IS the B operand of the next layer with no data
//     movement: for k-step t of an input row block, lane half h supplies the row that already
//     sits in its accumulator register t (rows (t&3)+8(t>>2)+4h).  The weights are pre-packed
//     in that k order (pda_sa_mlp_pack_weights), so hidden activations never leave registers;
//   * the folded BN + ReLU epilogue is applied in registers; the last layer is produced
//     128 output rows at a time and max-pooled over nsample with DPP row reductions across
//     the lanes of a group; groups wider than 32 samples are combined across waves through
//     LDS, which also stages the pooled tile for the store;
//   * weights stream from L2 (256 B fragments, the 4 waves of a workgroup read the same
//     fragment), packed four k-steps per 16-byte lane load.
// FLOPs per scale: 2 * M * ns * (c0 c1 + c1 c2 + c2 c3)  (BASELINE.md section 2).
#include "pda_common.h"

namespace pda {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SA_WAVES = 4;
constexpr int SA_CH = 4;  // output row blocks (of 32 rows) accumulated at a time

// A-fragment order: element e of the float4 at [(tq * R + rb) * 64 + lane] is
// W[rb*32 + (lane&31)][k(4*tq + e, lane>>5)].
//   mode 0 (first layer, input gathered):   k(t, h) = 2 t + h
//   mode 1 (input = previous accumulator):  k(t, h) = 32 (t / 16) + ((t%16)&3) + 8 ((t%16)>>2) + 4 h
//   mode 2 (input = rows of a (tokens, K) matrix read as float4 chunks): k(t, h) = 8 (t / 4) + 4 h + (t % 4)
// trans: the source is the TRANSPOSE of the matrix to pack (w is (cols, rows) row-major): input-gradient GEMMs.
__global__ void sa_mlp_pack_kernel(const float* __restrict__ w, float* __restrict__ wf, int rows, int cols,
                                   int R, int KS, int mode, int trans = 0) {
    const int total = (KS / 4) * R * 64 * 4;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
        const int e = o & 3, lane = (o >> 2) & 63, rest = o >> 8;
        const int rb = rest % R, tq = rest / R;
        const int t = tq * 4 + e, h = lane >> 5;
        const int row = rb * 32 + (lane & 31);
        int k;
        if (mode == 0) k = 2 * t + h;
        else if (mode == 2) k = 8 * (t >> 2) + 4 * h + (t & 3);
        else { const int tt = t & 15; k = 32 * (t >> 4) + (tt & 3) + 8 * (tt >> 2) + 4 * h; }
        wf[o] = (row < rows && k < cols) ? (trans ? w[(size_t)k * rows + row] : w[(size_t)row * cols + k]) : 0.f;
    }
}

// max over groups of NS consecutive lanes inside each 32-lane half (NS a power of two <= 32).
// The maximum of a group ends up in its LAST lane.
template <int NS>
__device__ __forceinline__ float group_max(float v) {
    if (NS >= 2) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));
    if (NS >= 4) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(v));
    if (NS >= 8) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(v));
    if (NS >= 16) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(v));
    if (NS >= 32) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    return v;
}

struct SaMlpParams {
    const float* xyz;       // (b, n, 3)
    const float* new_xyz;   // (b, m, 3)
    const float* feat;      // (b, c, n) or null
    const int32_t* idx;     // (b, m, ns)
    const float* wf[3];     // packed weights
    const float* scale[3];  // folded BN: y = relu(scale * (W x) + shift), padded to 32 R_l
    const float* shift[3];
    float* out;             // (b, c_out, m)
    int n, m, c, ns, c_out;
};

// acc[r] += W[rows of block rb0 + r] x Hin : one chunk of CH output row blocks over all k-steps.
// Hin is indexed statically: register t of element (t / 16) when FROM_ACC, else X0[t].
template <int KS, int R, int CH, typename HinT>
__device__ __forceinline__ void gemm_chunk(f32x16 (&acc)[CH], const HinT& hin, const float* __restrict__ wf,
                                            int rb0, int lane) {
    // One "step" = U float4 fragment loads per row block = 4*U k-steps = CH*4*U >= 16 MFMAs
    // (>= 1024 cycles).  The fragments of step s+1 are requested BEFORE the MFMAs of step s and a
    // sched_barrier keeps hipcc from sinking the loads back next to their use: with one wave per
    // SIMD nothing else hides the L2 latency of the weight stream (without this the MFMA pipe
    // waited ~1/3 of the time on s_waitcnt vmcnt, profiles/r01_sa_mlp_fused_microbench.txt).
    constexpr int U = CH >= 4 ? 1 : (CH == 2 ? 2 : 4);
    constexpr int TQ = KS / 4;
    constexpr int STEPS = (TQ + U - 1) / U;
#pragma unroll
    for (int r = 0; r < CH; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    float4 cur[CH][U], nxt[CH][U];
    auto load = [&](float4 (&dst)[CH][U], int step) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const int tq = step * U + u;
                if (tq < TQ)
                    dst[r][u] = *reinterpret_cast<const float4*>(wf + ((size_t)(tq * R + rb0 + r) * 64 + lane) * 4);
            }
    };
    load(cur, 0);
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        if (step + 1 < STEPS) load(nxt, step + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tq = step * U + u;
            if (tq < TQ) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float b = hin(tq * 4 + e);
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        const float4 a4 = cur[r][u];
                        const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[r], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < CH; ++r) cur[r][u] = nxt[r][u];
    }
}

// gemm_chunk with a RUN-TIME number of row blocks R (training-form kernels: one instantiation per K serves every width);
// the k loop stays fully unrolled -- the B operand is a register array and must be indexed statically.
template <int KS, int CH, typename HinT>
__device__ __forceinline__ void gemm_chunk_rt(f32x16 (&acc)[CH], const HinT& hin, const float* __restrict__ wf, int R,
                                               int rb0, int lane) {
    constexpr int TQ = KS / 4;
#pragma unroll
    for (int r = 0; r < CH; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    // weight fragments of k-chunks tq+1 and tq+2 are in flight while chunk tq feeds the MFMAs (one wave per SIMD:
    // nothing else hides the L2 latency of the weight stream)
    float4 buf[3][CH];
    const float* base = wf + ((size_t)rb0 * 64 + lane) * 4;
    const size_t step = (size_t)R * 64 * 4;
#pragma unroll
    for (int r = 0; r < CH; ++r) buf[0][r] = *reinterpret_cast<const float4*>(base + (size_t)r * 256);
    if (TQ > 1) {
#pragma unroll
        for (int r = 0; r < CH; ++r) buf[1][r] = *reinterpret_cast<const float4*>(base + step + (size_t)r * 256);
    }
#pragma unroll
    for (int tq = 0; tq < TQ; ++tq) {
        if (tq + 2 < TQ) {
#pragma unroll
            for (int r = 0; r < CH; ++r) buf[(tq + 2) % 3][r] = *reinterpret_cast<const float4*>(base + (size_t)(tq + 2) * step + (size_t)r * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float b = hin(tq * 4 + e);
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                const float4 a4 = buf[tq % 3][r];
                const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[r], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// folded BN + ReLU on one 32-row block held in accumulator layout
__device__ __forceinline__ void bn_relu(f32x16& v, const float* __restrict__ scale,
                                         const float* __restrict__ shift, int rb, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 s = *reinterpret_cast<const float4*>(scale + rb * 32 + 8 * q + 4 * h);
        const float4 t = *reinterpret_cast<const float4*>(shift + rb * 32 + 8 * q + 4 * h);
        v[4 * q + 0] = fmaxf(__builtin_fmaf(v[4 * q + 0], s.x, t.x), 0.f);
        v[4 * q + 1] = fmaxf(__builtin_fmaf(v[4 * q + 1], s.y, t.y), 0.f);
        v[4 * q + 2] = fmaxf(__builtin_fmaf(v[4 * q + 2], s.z, t.z), 0.f);
        v[4 * q + 3] = fmaxf(__builtin_fmaf(v[4 * q + 3], s.w, t.w), 0.f);
    }
}

template <int KS0>
struct X0Reader {
    const float (&x)[KS0];
    __device__ __forceinline__ float operator()(int t) const { return x[t]; }
};
template <int RIN>
struct AccReader {
    const f32x16 (&hv)[RIN];
    __device__ __forceinline__ float operator()(int t) const { return hv[t >> 4][t & 15]; }
};

// KS0: k-steps of layer 1 (= roundup4(ceil((3 + c) / 2))); R1, R2, R3: 32-row blocks per layer.
// NSL = min(ns, 32): lanes per group inside a 32-column block.
template <int KS0, int R1, int R2, int R3, int NSL>
__global__ __launch_bounds__(SA_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void sa_mlp_kernel(const SaMlpParams p) {
    extern __shared__ float pool[];  // [SA_WAVES][32 / NSL][R3 * 32]
    const int w = wave_id();
    const int lane = lane_id();
    const int h = lane >> 5, j = lane & 31;
    const int bs = blockIdx.y;
    const int ns = p.ns;
    const int gpb = 32 / NSL;                    // groups per 32-column block
    const int bpg = ns > 32 ? ns / 32 : 1;       // blocks per group
    // workgroup covers SA_WAVES blocks = SA_WAVES*32 consecutive columns of the (m*ns) axis
    const int64_t col0 = ((int64_t)blockIdx.x * SA_WAVES + w) * 32;
    const int64_t ncols = (int64_t)p.m * ns;
    const int64_t col = col0 + j;
    const bool colvalid = col < ncols;
    const int centre = colvalid ? (int)(col / ns) : 0;

    // ---- gather the grouped input of my column into B-operand registers ----------------
    const int id = colvalid ? p.idx[(size_t)bs * ncols + col] : 0;
    float x0[KS0];
    {
        const float* pt = p.xyz + ((size_t)bs * p.n + id) * 3;
        const float* ct = p.new_xyz + ((size_t)bs * p.m + centre) * 3;
        const float dx = pt[0] - ct[0], dy = pt[1] - ct[1], dz = pt[2] - ct[2];  // pointnet2_utils.py:692
        x0[0] = h == 0 ? dx : dy;  // k = 0 | 1
        const float* f = p.feat ? p.feat + (size_t)bs * p.c * p.n + id : nullptr;
        // k-step 1: k = 2 (dz) | 3 (feature 0)
        x0[1] = h == 0 ? dz : ((f && 0 < p.c) ? f[0] : 0.f);
#pragma unroll
        for (int t = 2; t < KS0; ++t) {
            const int ch = 2 * t + h - 3;
            x0[t] = (f && ch < p.c) ? f[(size_t)ch * p.n] : 0.f;
        }
    }

    // ---- layer 1 and 2: activations stay in accumulator registers ------------------------
    f32x16 h1[R1];
    {
        const X0Reader<KS0> rd{x0};
#pragma unroll
        for (int rb0 = 0; rb0 < R1; rb0 += (R1 < SA_CH ? R1 : SA_CH)) {
            constexpr int CH = R1 < SA_CH ? R1 : SA_CH;
            f32x16 acc[CH];
            gemm_chunk<KS0, R1, CH>(acc, rd, p.wf[0], rb0, lane);
#pragma unroll
            for (int r = 0; r < CH; ++r) { bn_relu(acc[r], p.scale[0], p.shift[0], rb0 + r, h); h1[rb0 + r] = acc[r]; }
        }
    }
    f32x16 h2[R2];
    {
        const AccReader<R1> rd{h1};
        // h1 + h2 + the chunk accumulators must fit the 512-register file: 2 blocks at a time
        // when h2 alone is 256 registers
        constexpr int CH2 = R2 >= 16 ? 2 : (R2 < SA_CH ? R2 : SA_CH);
#pragma unroll
        for (int rb0 = 0; rb0 < R2; rb0 += CH2) {
            constexpr int CH = CH2;
            f32x16 acc[CH];
            gemm_chunk<R1 * 16, R2, CH>(acc, rd, p.wf[1], rb0, lane);
#pragma unroll
            for (int r = 0; r < CH; ++r) { bn_relu(acc[r], p.scale[1], p.shift[1], rb0 + r, h); h2[rb0 + r] = acc[r]; }
        }
    }
    // ---- layer 3 in chunks, pooled over the lanes of each group --------------------------
    {
        const AccReader<R2> rd{h2};
        float* mypool = pool + (size_t)w * gpb * (R3 * 32);
#pragma unroll
        for (int rb0 = 0; rb0 < R3; rb0 += (R3 < SA_CH ? R3 : SA_CH)) {
            constexpr int CH = R3 < SA_CH ? R3 : SA_CH;
            f32x16 acc[CH];
            gemm_chunk<R2 * 16, R3, CH>(acc, rd, p.wf[2], rb0, lane);
#pragma unroll
            for (int r = 0; r < CH; ++r) {
                bn_relu(acc[r], p.scale[2], p.shift[2], rb0 + r, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float mx = group_max<NSL>(acc[r][i]);
                    if ((j & (NSL - 1)) == NSL - 1) {  // last lane of a group holds its maximum
                        const int row = (rb0 + r) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        mypool[(j / NSL) * (R3 * 32) + row] = mx;
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- combine blocks of wide groups and store (b, c_out, m) ---------------------------
    // the workgroup's SA_WAVES*32 columns hold G = SA_WAVES*gpb/bpg whole groups = consecutive centres
    const int G = SA_WAVES * gpb / bpg;
    const int64_t centre0 = ((int64_t)blockIdx.x * SA_WAVES * 32) / ns;
    for (int e = threadIdx.x; e < p.c_out * G; e += blockDim.x) {
        const int g = e % G, row = e / G;
        const int64_t cen = centre0 + g;
        if (cen >= p.m) continue;
        float v;
        if (bpg == 1) {
            v = pool[(size_t)g * (R3 * 32) + row];  // (w, group-in-block) pairs are laid out as g
        } else {
            v = pool[(size_t)(g * bpg) * gpb * (R3 * 32) + row];
            for (int q = 1; q < bpg; ++q) v = fmaxf(v, pool[(size_t)(g * bpg + q) * gpb * (R3 * 32) + row]);
        }
        p.out[((size_t)bs * p.c_out + row) * p.m + cen] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Training form (round 2): the group MLP's GEMMs one layer at a time on the same MFMA machinery.
// Training-mode BatchNorm needs the statistics of a layer's pre-activation before the next layer can start, so the
// chain cannot stay in registers (DESIGN.md "The vanilla SA group MLP in training mode"); what can be done is to run
// every contraction of the chain -- forward and input gradient -- on this file's f32 MFMA code instead of a library:
//   lin_cols_kernel<KS, GATHER = false>:  Y (T, N) = X (T, K) W^T, X rows read as float4 chunks straight into the
//       B-operand registers (k order "mode 2"), weights pre-packed in that order (also from W^T for dX = dY W);
//   lin_cols_kernel<KS0, GATHER = true>:  layer 1 with the grouping fused in: X is never built, the wave gathers
//       [xyz[idx] - centre | features[idx]] of its 32 (centre, sample) columns into registers as sa_mlp_kernel does.
// A wave owns 32 tokens and streams the whole packed weight matrix from L2 (4 waves of a workgroup read the same
// fragments); output rows leave as 16-byte stores.  The statistics, normalisation, ReLU and max-pool stay in
// csrc/bn_relu.hip, the weight gradient in csrc/wgrad.hip.
struct LinColsParams {
    const float* x;         // (T, K) row-major                         (GATHER: unused)
    const float* wf;        // packed weights
    float* y;               // (T, N) row-major
    int64_t tokens;
    int k, n_out, R;        // R = ceil(n_out / 32)
    // GATHER
    const float* xyz; const float* new_xyz; const float* feat_pm; const int32_t* idx;
    int n, m, c, ns;
};

// Up to two waves per SIMD where the registers allow it (K = 256: 166 VGPRs): one wave's strided operand loads and
// output stores then run under the other's MFMA chain.
template <int KS, bool GATHER>
__global__ __launch_bounds__(SA_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 2)))
void lin_cols_kernel(const LinColsParams p) {
    const int w = wave_id();
    const int lane = lane_id();
    const int h = lane >> 5, j = lane & 31;
    const int64_t tok = ((int64_t)blockIdx.x * SA_WAVES + w) * 32 + j;
    const bool valid = tok < p.tokens;
    const int64_t tk = valid ? tok : 0;
    float x[KS];
    if constexpr (!GATHER) {
        const float* row = p.x + tk * p.k + 4 * h;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(row + 8 * q);
            x[4 * q] = valid ? v.x : 0.f; x[4 * q + 1] = valid ? v.y : 0.f; x[4 * q + 2] = valid ? v.z : 0.f; x[4 * q + 3] = valid ? v.w : 0.f;
        }
    } else {
        // token = (scene, centre, sample); k order of mode 0: k = 2 t + h; k = 0..2 centred xyz, k = 3 + ch features
        const int64_t per_scene = (int64_t)p.m * p.ns;
        const int bs = (int)(tk / per_scene);
        const int centre = (int)((tk - bs * per_scene) / p.ns);
        const int id = p.idx[tk];
        const float* pt = p.xyz + ((size_t)bs * p.n + id) * 3;
        const float* ct = p.new_xyz + ((size_t)bs * p.m + centre) * 3;
        const float dx = pt[0] - ct[0], dy = pt[1] - ct[1], dz = pt[2] - ct[2];     // pointnet2_utils.py:692
        const float* f = p.feat_pm + ((size_t)bs * p.n + id) * p.c;                 // point-major row of the neighbour
        x[0] = valid ? (h == 0 ? dx : dy) : 0.f;
        x[1] = valid ? (h == 0 ? dz : (0 < p.c ? f[0] : 0.f)) : 0.f;
#pragma unroll
        for (int t = 2; t < KS; ++t) {
            const int ch = 2 * t + h - 3;
            x[t] = (valid && ch < p.c) ? f[ch] : 0.f;
        }
    }
    const X0Reader<KS> rd{x};
    float* yrow = p.y + tk * p.n_out;
    for (int rb0 = 0; rb0 < p.R; rb0 += SA_CH) {
        f32x16 acc[SA_CH];
        gemm_chunk_rt<KS, SA_CH>(acc, rd, p.wf, p.R, rb0, lane);    // n_out % 128 == 0 (launcher): whole chunks only
        if (valid) {
#pragma unroll
            for (int r = 0; r < SA_CH; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(yrow + (rb0 + r) * 32 + 8 * q + 4 * h) =
                        make_float4(acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]);
        }
    }
}

template <int KS0, int R1, int R2, int R3>
static int launch_sa_mlp(const SaMlpParams& p, int b, hipStream_t stream) {
    const int ns = p.ns;
    const int nsl = ns < 32 ? ns : 32;
    const int gpb = 32 / nsl;
    const size_t lds = (size_t)SA_WAVES * gpb * R3 * 32 * sizeof(float);
    const int64_t ncols = (int64_t)p.m * ns;
    dim3 grid((unsigned)divup64(ncols, SA_WAVES * 32), b), block(SA_WAVES * 64);
    void (*kern)(const SaMlpParams) = nullptr;
    switch (nsl) {
        case 32: kern = sa_mlp_kernel<KS0, R1, R2, R3, 32>; break;
        case 16: kern = sa_mlp_kernel<KS0, R1, R2, R3, 16>; break;
        case 8: kern = sa_mlp_kernel<KS0, R1, R2, R3, 8>; break;
        default: return PDA_ERR_UNSUPPORTED;
    }
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PDA_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    return check_launch("pda_sa_mlp_maxpool");
}

}  // namespace pda

PDA_API int pda_sa_mlp_packed_size(int rows, int cols, int first_layer) {
    if (rows <= 0 || cols <= 0) return 0;
    const int R = pda::divup(rows, 32);
    const int KS = first_layer ? pda::divup(pda::divup(cols, 2), 4) * 4 : pda::divup(cols, 32) * 16;
    return KS * R * 64;  // floats
}

PDA_API int pda_sa_mlp_pack_weights(const float* w, float* wf, int rows, int cols, int first_layer,
                                    pda_stream_t stream) {
    PDA_REQUIRE(w && wf && rows > 0 && cols > 0, "pda_sa_mlp_pack_weights: bad argument");
    const int R = pda::divup(rows, 32);
    const int KS = first_layer ? pda::divup(pda::divup(cols, 2), 4) * 4 : pda::divup(cols, 32) * 16;
    const int total = KS * R * 64;
    hipLaunchKernelGGL(pda::sa_mlp_pack_kernel, dim3(pda::divup(total, 256) < 1024 ? pda::divup(total, 256) : 1024),
                       dim3(256), 0, (hipStream_t)stream, w, wf, rows, cols, R, KS, first_layer ? 0 : 1);
    return pda::check_launch("pda_sa_mlp_pack_weights");
}

PDA_API int pda_sa_mlp_maxpool(const float* xyz, const float* new_xyz, const float* features,
                               const int32_t* idx, float* out, int b, int n, int m, int c, int ns,
                               const int32_t* dims, const float* const* wf, const float* const* scale,
                               const float* const* shift, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n > 0 && m >= 0 && c >= 0 && ns >= 1, "pda_sa_mlp_maxpool: bad size");
    if (b == 0 || m == 0) return PDA_OK;
    PDA_REQUIRE(xyz && new_xyz && idx && out && dims && wf && scale && shift, "pda_sa_mlp_maxpool: null pointer");
    PDA_REQUIRE(features || c == 0, "pda_sa_mlp_maxpool: c > 0 but features is null");
    PDA_REQUIRE(dims[0] == 3 + c, "pda_sa_mlp_maxpool: dims[0]=%d must be 3 + c = %d", dims[0], 3 + c);
    if ((ns & (ns - 1)) != 0 || ns < 8 || ns > 128 || b > 65535) {
        pda::set_error("pda_sa_mlp_maxpool: nsample=%d not a power of two in [8,128]", ns);
        return PDA_ERR_UNSUPPORTED;
    }
    pda::SaMlpParams p{};
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat = features; p.idx = idx; p.out = out;
    p.n = n; p.m = m; p.c = c; p.ns = ns; p.c_out = dims[3];
    for (int l = 0; l < 3; ++l) { p.wf[l] = wf[l]; p.scale[l] = scale[l]; p.shift[l] = shift[l]; }
    const int ks0 = pda::divup(pda::divup(dims[0], 2), 4) * 4;
    const int r1 = pda::divup(dims[1], 32), r2 = pda::divup(dims[2], 32), r3 = pda::divup(dims[3], 32);
    const hipStream_t s = (hipStream_t)stream;
#define PDA_SA_CASE(K, A, B, C) \
    if (ks0 == K && r1 == A && r2 == B && r3 == C) return pda::launch_sa_mlp<K, A, B, C>(p, b, s)
    PDA_SA_CASE(4, 1, 1, 1);      // 4 -> 16 -> 16 -> 32         (layer 0, scale 0)
    PDA_SA_CASE(4, 1, 1, 2);      // 4 -> 32 -> 32 -> 64         (layer 0, scale 1)
    PDA_SA_CASE(132, 8, 8, 16);   // 259 -> 256 -> 256 -> 512    (layer 5)
    PDA_SA_CASE(132, 8, 16, 16);  // 259 -> 256 -> 512 -> 512    (ONCE layer 5, scale 2)
    PDA_SA_CASE(132, 8, 16, 32);  // 259 -> 256 -> 512 -> 1024   (KITTI layer 5, scale 1)
#undef PDA_SA_CASE
    pda::set_error("pda_sa_mlp_maxpool: no kernel built for chain %d->%d->%d->%d", dims[0], dims[1], dims[2], dims[3]);
    return PDA_ERR_UNSUPPORTED;
}

// ---- training form: one GEMM of the group MLP per call (see lin_cols_kernel) ----------------------------------------
PDA_API int pda_linear_cols_packed_size(int n_out, int k) {
    if (n_out <= 0 || k <= 0) return 0;
    return pda::divup(k, 8) * 4 * pda::divup(n_out, 32) * 64;   // KS * R * 64 floats, KS = k / 2 rounded to 4 k-steps
}

PDA_API int pda_linear_cols_pack(const float* w, float* wf, int n_out, int k, int transposed_source, int gather_order,
                                 pda_stream_t stream) {
    PDA_REQUIRE(w && wf && n_out > 0 && k > 0, "pda_linear_cols_pack: bad argument");
    const int R = pda::divup(n_out, 32);
    const int KS = pda::divup(k, 8) * 4;
    const int total = KS * R * 64;
    hipLaunchKernelGGL(pda::sa_mlp_pack_kernel, dim3(pda::divup(total, 256) < 1024 ? pda::divup(total, 256) : 1024), dim3(256), 0,
                       (hipStream_t)stream, w, wf, n_out, k, R, KS, gather_order ? 0 : 2, transposed_source);
    return pda::check_launch("pda_linear_cols_pack");
}

PDA_API int pda_linear_cols(const float* x, const float* wf, float* y, int64_t tokens, int k, int n_out, pda_stream_t stream) {
    PDA_REQUIRE(tokens >= 0 && k > 0 && n_out > 0, "pda_linear_cols: bad size");
    if (tokens == 0) return PDA_OK;
    PDA_REQUIRE(x && wf && y && (((uintptr_t)x | (uintptr_t)wf | (uintptr_t)y) & 15) == 0, "pda_linear_cols: null or misaligned pointer");
    if ((k != 256 && k != 512) || n_out % 128 != 0 || n_out > 1024) {
        pda::set_error("pda_linear_cols: no kernel built for K=%d, N=%d (K in {256, 512}, N a multiple of 128 <= 1024)", k, n_out);
        return PDA_ERR_UNSUPPORTED;
    }
    pda::LinColsParams p{};
    p.x = x; p.wf = wf; p.y = y; p.tokens = tokens; p.k = k; p.n_out = n_out; p.R = n_out / 32;
    const dim3 grid((unsigned)pda::divup64(tokens, pda::SA_WAVES * 32)), block(pda::SA_WAVES * 64);
    if (k == 256) hipLaunchKernelGGL((pda::lin_cols_kernel<128, false>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((pda::lin_cols_kernel<256, false>), grid, block, 0, (hipStream_t)stream, p);
    return pda::check_launch("pda_linear_cols");
}

PDA_API int pda_sa_gather_linear(const float* xyz, const float* new_xyz, const float* feats_pm, const int32_t* idx, const float* wf,
                                 float* y, int b, int n, int m, int c, int ns, int n_out, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n > 0 && m >= 0 && c >= 0 && ns >= 1 && n_out > 0, "pda_sa_gather_linear: bad size");
    const int64_t tokens = (int64_t)b * m * ns;
    if (tokens == 0) return PDA_OK;
    PDA_REQUIRE(xyz && new_xyz && idx && wf && y && (feats_pm || c == 0), "pda_sa_gather_linear: null pointer");
    PDA_REQUIRE((((uintptr_t)wf | (uintptr_t)y) & 15) == 0, "pda_sa_gather_linear: misaligned pointer");
    if (c != 256 || n_out % 128 != 0 || n_out > 1024) {
        pda::set_error("pda_sa_gather_linear: no kernel built for C=%d, N=%d (C = 256, N a multiple of 128 <= 1024)", c, n_out);
        return PDA_ERR_UNSUPPORTED;
    }
    pda::LinColsParams p{};
    p.wf = wf; p.y = y; p.tokens = tokens; p.k = 3 + c; p.n_out = n_out; p.R = n_out / 32;
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat_pm = feats_pm; p.idx = idx; p.n = n; p.m = m; p.c = c; p.ns = ns;
    const dim3 grid((unsigned)pda::divup64(tokens, pda::SA_WAVES * 32)), block(pda::SA_WAVES * 64);
    hipLaunchKernelGGL((pda::lin_cols_kernel<132, true>), grid, block, 0, (hipStream_t)stream, p);
    return pda::check_launch("pda_sa_gather_linear");
}
