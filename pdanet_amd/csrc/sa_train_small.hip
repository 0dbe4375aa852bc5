// sa_train_small.hip -- the NARROW vanilla set-abstraction scale in TRAINING form, forward and backward, as a family
// of recompute passes (ONCE / KITTI layer 0: 4 -> 16 -> 16 -> 32 over 16 neighbours, 4 -> 32 -> 32 -> 64 over 32).
//
// Reference chain (pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py:1657-1670, pointnet2_utils.py:671-704):
//   QueryAndGroup -> [Conv2d 1x1 (no bias) -> BatchNorm2d (batch statistics) -> ReLU] x 3 -> max over nsample.
// Layer by layer that materialises five (B, M, ns, C) tensors per scale (0.13 - 0.27 GB each at 2 x 16384 centres) and
// moves them ~20 times through HBM for 7.6 GFLOP of arithmetic.  This regime is the opposite of layer 5's (DESIGN.md
// "The vanilla SA group MLP in training mode"): recomputing a token from its gathered 16-byte input costs 0.8 - 3.2 k
// MACs, re-reading its activations costs 0.4 - 1 KB.  So nothing between the neighbour lists and the pooled output
// exists in HBM in the forward pass, and the backward pass keeps two narrow gradient tensors only:
//
//   forward   F1  x0 -> z1                          per-channel sum / sum of squares of z1
//             F2  x0 -> z1 -> y1 -> z2              ... of z2            (y = relu(batchnorm(z)), batch statistics)
//             F3  x0 -> ... -> y2 -> z3             ... of z3
//             F4  x0 -> ... -> z3 -> y3 -> max      out (B, M, C3), the pre-activation at the arg-max, the arg-max slot
//   backward  E0  (out-side only)                   BatchNorm-3 backward sums from (grad_out, zmax, arg)
//             B3  x0 -> ... -> z3, dz3              dW3 += dz3 y2^T, dy2 = W3^T dz3, dz2' = dy2 [y2 > 0] -> HBM, BN-2 sums
//             B2  x0 -> z1 -> y1 -> z2, dz2         dW2 += dz2 y1^T, dy1 = W2^T dz2, dz1' -> HBM, BN-1 sums
//             B1  x0 -> z1, dz1                     dW1 += dz1 x0^T
// with a one-workgroup finalize launch behind every pass (fixed-order double sums of the per-workgroup partials: the
// statistics, running_mean / running_var, dgamma / dbeta and the weight gradients are deterministic).
//
// Arithmetic: exact f32 on v_mfma_f32_32x32x2_f32 with the machinery of csrc/sa_mlp.hip -- a wave owns a tile of 32
// tokens (lane & 31), the 32x32 accumulator of a layer (rows = channels, columns = tokens) IS the B operand of the next
// layer, forward (W x H) and backward (W^T x dZ) alike, so activations and gradients never leave registers between
// layers.  Execution shape (what the counters asked for, tools/pmc_ss.sh): per tile the MFMA chain (<= 116 instructions),
// ~1200 VALU instructions of BatchNorm / ReLU / gradient algebra and the LDS transposes are ONE dependent sequence, so a
// workgroup is 8 waves = two per SIMD and one wave's VALU / memory phases run under the other's MFMA chain.  To fit 256
// registers per wave the packed weight fragments and the per-channel state (mean, invstd, gamma, beta, backward means) live
// in LDS, one copy per workgroup (re-read per tile: a compiler barrier keeps them from being hoisted into registers), the
// gather runs two tiles ahead, and B3 walks the 32-channel blocks of layer 3 one at a time (same summation order).  The
// weight gradient contracts over tokens: the tile's dZ and Y blocks are transposed through a wave-private LDS image (row
// stride 33, no barriers) into operand layout.  The arg-max runs on the transposed z3 tile with one lane per (group,
// channel): strict '>' over the slots in order = lowest slot on ties (ball query pads short lists with repeats of the first
// neighbour, so ties are the rule).
#include "pda_common.h"

namespace pda {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SS_WAVES = 8;                         // waves per workgroup: two per SIMD -- one wave's VALU / memory work runs under the other's MFMA chain
constexpr int SS_LD = 33;                           // row stride (floats) of the transposed tiles
constexpr int SS_LDS_WAVE = 96 * SS_LD;             // per wave: a 64-row block + a 32-row block
constexpr int SS_LDS_WPACK = 6400;                   // LDS (floats): [packed weights | state | per-wave tiles]
constexpr int SS_LDS_STATE = 768;
__host__ __device__ constexpr int ss_lds_floats(int waves) { return SS_LDS_WPACK + SS_LDS_STATE + waves * SS_LDS_WAVE; }
constexpr int SS_MAX_BLOCKS = 512;                  // workgroups per pass = per-pass partials
constexpr int SS_PART = 128;                        // doubles per workgroup partial: [2][64]
constexpr int SS_DW = 2048;                         // floats per workgroup weight-gradient partial: [64][32]
// state (floats): per layer l = 1..3 with P = 32 (layers 1, 2) or 32 R3 (layer 3): mean[P] istd[P] gamma[P] beta[P] m1[P] m2[P]
__host__ __device__ constexpr int ss_state_base(int layer) { return (layer - 1) * 192; }
static_assert(384 + 6 * 64 <= 1024, "the state (floats) fits the first 4 KB of the workspace");
// packed weight fragments (floats): [W1 | W2 | W3 | W2^T | W3^T]
__host__ __device__ constexpr int ss_off_w2() { return 256; }
__host__ __device__ constexpr int ss_off_w3() { return 256 + 1024; }
__host__ __device__ constexpr int ss_off_w2t(int R3) { return 256 + 1024 + 1024 * R3; }
__host__ __device__ constexpr int ss_off_w3t(int R3) { return 256 + 2048 + 1024 * R3; }
__host__ __device__ constexpr int ss_wpack_floats(int R3) { return 256 + 2048 + 2048 * R3; }

struct SsParams {
    const float* xyz;        // (b, n, 3)
    const float* new_xyz;    // (b, m, 3)
    const float* feat;       // (b, n, c) point-major (c == 1: the same memory as (b, 1, n)) or null
    const int32_t* idx;      // (b, m, ns)
    const float* wpack;
    const float* state;
    double* partial;         // [gridDim.x][2][64]
    float* out;              // F4: (b*m, c3)
    float* zmax;
    uint8_t* arg;
    const float* gout;       // backward: (b*m, c3)
    const uint8_t* arg_in;
    float* dz_out;           // B3: dz2' (tokens, c2); B2: dz1' (tokens, c1)
    const float* dz_in;      // B2: dz2'; B1: dz1'
    float* dw_partial;       // [gridDim.x][64][32]
    int n, m, c, ns, c1, c2, c3;
    int64_t tiles;           // tokens / 32
};

// ---- packing: A-operand fragments in the k order the consumer supplies its B operand in (cf. sa_mlp_pack_kernel) ----
//   mode 0 (B operand gathered):               k(t, h) = 2 t + h
//   mode 1 (B operand = a 32-row accumulator): k(t, h) = 32 (t / 16) + ((t % 16) & 3) + 8 ((t % 16) >> 2) + 4 h
__device__ __forceinline__ float ss_pack_one(const float* __restrict__ w, int o, int rows, int cols, int R, int mode, int trans) {
    const int e = o & 3, lane = (o >> 2) & 63, rest = o >> 8;
    const int rb = rest % R, tq = rest / R;
    const int t = tq * 4 + e, h = lane >> 5;
    const int row = rb * 32 + (lane & 31);
    int k;
    if (mode == 0) k = 2 * t + h;
    else { const int tt = t & 15; k = 32 * (t >> 4) + (tt & 3) + 8 * (tt >> 2) + 4 * h; }
    if (row >= rows || k >= cols) return 0.f;
    return trans ? w[(size_t)k * rows + row] : w[(size_t)row * cols + k];
}

__global__ void ss_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                               float* __restrict__ wpack, int c0, int c1, int c2, int c3, int R3) {
    const int total = ss_wpack_floats(R3);
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
        float v;
        if (o < ss_off_w2()) v = ss_pack_one(w1, o, c1, c0, 1, 0, 0);
        else if (o < ss_off_w3()) v = ss_pack_one(w2, o - ss_off_w2(), c2, c1, 1, 1, 0);
        else if (o < ss_off_w2t(R3)) v = ss_pack_one(w3, o - ss_off_w3(), c3, c2, R3, 1, 0);
        else if (o < ss_off_w3t(R3)) v = ss_pack_one(w2, o - ss_off_w2t(R3), c1, c2, 1, 1, 1);     // W2^T: (c1 x c2), source (c2 x c1)
        else v = ss_pack_one(w3, o - ss_off_w3t(R3), c2, c3, 1, 1, 1);                             // W3^T: (c2 x c3), source (c3 x c2)
        wpack[o] = v;
    }
}

// ---- register-resident weight fragments and the chain's GEMM ---------------------------------------------------------
template <int KS, int R>
struct SsFrag {      // fragments of one packed matrix in LDS (shared by the workgroup's waves): one ds_read_b128 per 4 MFMAs
    const float* wf;
    int lane;
    __device__ __forceinline__ void load(const float* lds_wf, int l) { wf = lds_wf; lane = l; }
    __device__ __forceinline__ float4 f(int tq, int r) const { return *reinterpret_cast<const float4*>(wf + ((tq * R + r) * 64 + lane) * 4); }
};

// acc[r] = W[row block r] x Hin over KS k-steps; hin(t) = the lane's B operand of k-step t (static index)
template <int KS, int R, typename HinT>
__device__ __forceinline__ void ss_mm(f32x16 (&acc)[R], const HinT& hin, const SsFrag<KS, R>& w) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
#pragma unroll
    for (int tq = 0; tq < KS / 4; ++tq)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float b = hin(tq * 4 + e);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 a4 = w.f(tq, r);
                const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[r], 0, 0, 0);
            }
        }
}

// one output row block rb of W x Hin (the fragments of block rb only)
template <int KS, int R, typename HinT>
__device__ __forceinline__ void ss_mm_block(f32x16& acc, const HinT& hin, const SsFrag<KS, R>& w, int rb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int tq = 0; tq < KS / 4; ++tq) {
        const float4 a4 = w.f(tq, rb);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w)), hin(tq * 4 + e), acc, 0, 0, 0);
    }
}
// acc += W[:, k-steps 16 kb .. 16 kb + 15] x (one 32-row accumulator block): a slice of a longer contraction, in its order
template <int KS, typename HinT>
__device__ __forceinline__ void ss_mm_slice(f32x16& acc, const HinT& hin, const SsFrag<KS, 1>& w, int kb) {
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
        const float4 a4 = w.f(4 * kb + tq, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w)), hin(tq * 4 + e), acc, 0, 0, 0);
    }
}

struct SsX0 {
    const float (&x)[4];
    __device__ __forceinline__ float operator()(int t) const { return x[t]; }
};
template <int RIN>
struct SsAcc {
    const f32x16 (&hv)[RIN];
    __device__ __forceinline__ float operator()(int t) const { return hv[t >> 4][t & 15]; }
};

// per-channel constants of accumulator registers 4q .. 4q+3 of row block rb (channels rb*32 + 8q + 4h + 0..3)
struct SsChan { float4 mu, is, ga, be; };
__device__ __forceinline__ SsChan ss_chan(const float* __restrict__ st, int P, int rb, int q, int h) {
    const float* b = st + rb * 32 + 8 * q + 4 * h;
    SsChan c;
    c.mu = *reinterpret_cast<const float4*>(b);
    c.is = *reinterpret_cast<const float4*>(b + P);
    c.ga = *reinterpret_cast<const float4*>(b + 2 * P);
    c.be = *reinterpret_cast<const float4*>(b + 3 * P);
    return c;
}
__device__ __forceinline__ float f4(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// z (pre-activation block) -> xh = (z - mean) * invstd, y = relu(xh * gamma + beta): the expression of csrc/bn_relu.hip
template <bool KEEP_XH>
__device__ __forceinline__ void ss_bn_relu(f32x16& z, f32x16& xh, const float* __restrict__ st, int P, int rb, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const SsChan c = ss_chan(st, P, rb, q, h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = (z[4 * q + e] - f4(c.mu, e)) * f4(c.is, e);
            if (KEEP_XH) xh[4 * q + e] = x;
            z[4 * q + e] = fmaxf(x * f4(c.ga, e) + f4(c.be, e), 0.f);
        }
    }
}

// gather the grouped input [xyz[idx] - centre | features[idx]] of token `tok` (neighbour `id`) in B-operand layout (k = 2 t + h)
__device__ __forceinline__ void ss_gather(const SsParams& p, int64_t tok, int id, int h, float (&x0)[4]) {
    const int64_t per_scene = (int64_t)p.m * p.ns;
    const int bs = (int)(tok / per_scene);
    const int centre = (int)((tok - bs * per_scene) / p.ns);
    const float* pt = p.xyz + ((size_t)bs * p.n + id) * 3;
    const float* ct = p.new_xyz + ((size_t)bs * p.m + centre) * 3;
    const float dx = pt[0] - ct[0], dy = pt[1] - ct[1], dz = pt[2] - ct[2];      // pointnet2_utils.py:692
    const float* f = p.feat ? p.feat + ((size_t)bs * p.n + id) * p.c : nullptr;
    x0[0] = h == 0 ? dx : dy;
    x0[1] = h == 0 ? dz : ((f && 0 < p.c) ? f[0] : 0.f);
#pragma unroll
    for (int t = 2; t < 4; ++t) {
        const int ch = 2 * t + h - 3;
        x0[t] = (f && ch < p.c) ? f[ch] : 0.f;
    }
}

// The walk over a wave's tiles with the gather two deep in flight: a wave runs alone on its SIMD in the heavy passes, so
// nothing else hides the two dependent loads (neighbour index -> its coordinates) of a tile.  The index of tile i + 2 and
// the coordinates of tile i + 1 are requested before tile i's MFMA chain; clamped tiles read tile 0 and are never used.
struct SsWalk {
    int64_t tile, stride, tiles;
    int id1, id2;           // neighbour of my token in tile + stride, tile + 2 stride
    float xn[4];            // gathered input of my token in tile (ready), refilled for tile + stride by advance()
    __device__ __forceinline__ void start(const SsParams& p, int64_t first, int64_t step, int j, int h) {
        tile = first; stride = step; tiles = p.tiles;
        const int id0 = p.idx[(first < tiles ? first : 0) * 32 + j];
        id1 = p.idx[(first + step < tiles ? first + step : 0) * 32 + j];
        id2 = p.idx[(first + 2 * step < tiles ? first + 2 * step : 0) * 32 + j];
        ss_gather(p, (first < tiles ? first : 0) * 32 + j, id0, h, xn);
    }
    // x0 = this tile's input; requests the next tile's input and the index two tiles on
    __device__ __forceinline__ void fetch(const SsParams& p, int j, int h, float (&x0)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) x0[t] = xn[t];
        const int64_t n1 = tile + stride < tiles ? tile + stride : 0, n3 = tile + 3 * stride < tiles ? tile + 3 * stride : 0;
        ss_gather(p, n1 * 32 + j, id1, h, xn);
        id1 = id2;
        id2 = p.idx[n3 * 32 + j];
    }
};

__device__ __forceinline__ float ss_half_sum(float v) {     // sum over the 32 lanes of my half (every lane gets it)
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// per-lane (sum, weighted sum) registers of NB row blocks -> the workgroup's partial [2][64] (doubles, fixed order)
template <int NB, int W>
__device__ __forceinline__ void ss_store_partial(float (&s1)[NB][16], float (&s2)[NB][16], double* __restrict__ part, float* lds) {
    const int w = wave_id(), lane = lane_id(), h = lane >> 5;
    __syncthreads();                                   // the tile loop's LDS traffic is over in every wave
    for (int e = threadIdx.x; e < W * 128; e += blockDim.x) lds[e] = 0.f;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float a = ss_half_sum(s1[b][i]), c = ss_half_sum(s2[b][i]);
            if ((lane & 31) == 0) {
                const int ch = b * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                lds[w * 128 + ch] = a;
                lds[w * 128 + 64 + ch] = c;
            }
        }
    __syncthreads();
    if (threadIdx.x < 128) {
        double a = 0.0;
#pragma unroll
        for (int ww = 0; ww < W; ++ww) a += (double)lds[ww * 128 + threadIdx.x];
        part[threadIdx.x] = a;
    }
}

// the wave's weight-gradient accumulators (rows = output channels, lane & 31 = input channel) -> workgroup partial [64][32]
template <int NB, int W>
__device__ __forceinline__ void ss_store_dw(const f32x16 (&dw)[NB], float* __restrict__ part, float* lds) {
    const int w = wave_id(), lane = lane_id(), h = lane >> 5, j = lane & 31;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) lds[w * SS_DW + (b * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 32 + j] = dw[b][i];
    __syncthreads();
    for (int e = threadIdx.x; e < NB * 1024; e += blockDim.x) {
        float a = lds[e];
#pragma unroll
        for (int ww = 1; ww < W; ++ww) a += lds[ww * SS_DW + e];
        part[e] = a;
    }
}

// accumulator block (rows = channels, lane & 31 = token) -> transposed LDS rows [channel][token]
__device__ __forceinline__ void ss_to_lds(float* __restrict__ t, const f32x16& v, int rb, int lane) {
    const int h = lane >> 5, j = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) t[(rb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * SS_LD + j] = v[i];
}

// dW[rb] += Z[rb] Y^T over the tile's 32 tokens: A = Z^T rows (channel = lane & 31, token = 2 t + h), B likewise from Y
template <int NB>
__device__ __forceinline__ void ss_dw_tile(f32x16 (&dw)[NB], const float* __restrict__ zt, const float* __restrict__ yt, int lane) {
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        if ((t & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // at most four k-steps of operand reads in flight: registers
        const float b = yt[c * SS_LD + 2 * t + h];
#pragma unroll
        for (int rb = 0; rb < NB; ++rb)
            dw[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(zt[(rb * 32 + c) * SS_LD + 2 * t + h], b, dw[rb], 0, 0, 0);
    }
}

__device__ __forceinline__ void ss_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// packed weights and per-channel state -> LDS, once per workgroup
__device__ __forceinline__ void ss_stage_constants(const SsParams& p, float* smem, int R3) {
    const int nw = ss_wpack_floats(R3);
    for (int e = threadIdx.x * 4; e < nw; e += blockDim.x * 4)
        *reinterpret_cast<float4*>(smem + e) = *reinterpret_cast<const float4*>(p.wpack + e);
    for (int e = threadIdx.x * 4; e < SS_LDS_STATE; e += blockDim.x * 4)
        *reinterpret_cast<float4*>(smem + SS_LDS_WPACK + e) = *reinterpret_cast<const float4*>(p.state + e);
    __syncthreads();
}

// ---- forward passes ---------------------------------------------------------------------------------------------------
// STAGE 1..3: statistics of layer STAGE's pre-activation; STAGE 4: pooled output.  NS = nsample (16 | 32), R3 = c3 / 32.
template <int R3, int NS, int STAGE, int W>
__global__ __launch_bounds__(W * 64) void ss_fwd_kernel(const SsParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = wave_id(), lane = lane_id(), h = lane >> 5, j = lane & 31;
    float* lds_all = smem + SS_LDS_WPACK + SS_LDS_STATE;
    float* lds = lds_all + w * SS_LDS_WAVE;
    ss_stage_constants(p, smem, R3);
    SsFrag<4, 1> w1;
    SsFrag<16, 1> w2;
    SsFrag<16, R3> w3;
    w1.load(smem, lane);
    w2.load(smem + ss_off_w2(), lane);
    w3.load(smem + ss_off_w3(), lane);
    const float* st1 = smem + SS_LDS_WPACK + ss_state_base(1);
    const float* st2 = smem + SS_LDS_WPACK + ss_state_base(2);
    const float* st3 = smem + SS_LDS_WPACK + ss_state_base(3);
    constexpr int NB = STAGE == 3 ? R3 : 1;
    float s1[NB][16], s2[NB][16];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) { s1[b][i] = 0.f; s2[b][i] = 0.f; }
    // STAGE 4, entry role: lane e <-> (group e / c3 of the tile, channel e % c3); (32 / NS) * c3 == 64
    const int ech = lane % p.c3, eg = lane / p.c3;
    float emu = 0.f, eis = 0.f, ega = 0.f, ebe = 0.f;
    if (STAGE == 4) { emu = st3[ech]; eis = st3[32 * R3 + ech]; ega = st3[64 * R3 + ech]; ebe = st3[96 * R3 + ech]; }

    SsWalk walk;
    walk.start(p, (int64_t)blockIdx.x * W + w, (int64_t)gridDim.x * W, j, h);
    for (; walk.tile < p.tiles; walk.tile += walk.stride) {
        const int64_t tile = walk.tile;
        asm volatile("" ::: "memory");     // weights and per-channel constants are re-read from LDS per tile, not hoisted into registers
        float x0[4];
        walk.fetch(p, j, h, x0);
        f32x16 a1[1], a2[1], a3[R3], dummy;
        ss_mm<4, 1>(a1, SsX0{x0}, w1);
        if (STAGE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s1[0][i] += a1[0][i]; s2[0][i] += a1[0][i] * a1[0][i]; }
            continue;
        }
        ss_bn_relu<false>(a1[0], dummy, st1, 32, 0, h);
        ss_mm<16, 1>(a2, SsAcc<1>{a1}, w2);
        if (STAGE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s1[0][i] += a2[0][i]; s2[0][i] += a2[0][i] * a2[0][i]; }
            continue;
        }
        ss_bn_relu<false>(a2[0], dummy, st2, 32, 0, h);
        ss_mm<16, R3>(a3, SsAcc<1>{a2}, w3);
        if (STAGE == 3) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1[b][i] += a3[b][i]; s2[b][i] += a3[b][i] * a3[b][i]; }
            continue;
        }
        // ---- STAGE 4: transposed z3 tile, then one lane per (group, channel) scans the group's slots in order
#pragma unroll
        for (int rb = 0; rb < R3; ++rb) ss_to_lds(lds, a3[rb], rb, lane);
        ss_wave_fence();
        float best = -1.f, bz = 0.f;      // y >= 0: slot 0 always replaces the start value
        int bs = 0;
        const float* zrow = lds + ech * SS_LD + eg * NS;
#pragma unroll 8
        for (int s = 0; s < NS; ++s) {
            const float z = zrow[s];
            const float y = fmaxf((z - emu) * eis * ega + ebe, 0.f);
            if (y > best) { best = y; bz = z; bs = s; }
        }
        p.out[tile * 64 + lane] = best;
        p.zmax[tile * 64 + lane] = bz;
        p.arg[tile * 64 + lane] = (uint8_t)bs;
        ss_wave_fence();                  // my reads are done before the next tile's writes
    }
    if (STAGE <= 3) ss_store_partial<NB, W>(s1, s2, p.partial + (size_t)blockIdx.x * SS_PART, lds_all);
}

// ---- backward passes --------------------------------------------------------------------------------------------------
// E0: sums of dyh = gout [y3 > 0] and dyh * xh3 over the (group, channel) entries -- the only tokens the max-pool routes
// a gradient to.  thread = 4 consecutive channels of a group.
__global__ __launch_bounds__(256) void ss_pool_sums_kernel(const float* __restrict__ gout, const float* __restrict__ zmax,
                                                           const float* __restrict__ st3, int P3, int c3, int64_t groups,
                                                           double* __restrict__ partial) {
    __shared__ double red[2][256][4];
    const int cg = c3 / 4, col = threadIdx.x % cg, r0 = threadIdx.x / cg, rpb = 256 / cg;
    const float4 mu = *reinterpret_cast<const float4*>(st3 + 4 * col), is = *reinterpret_cast<const float4*>(st3 + P3 + 4 * col);
    const float4 ga = *reinterpret_cast<const float4*>(st3 + 2 * P3 + 4 * col), be = *reinterpret_cast<const float4*>(st3 + 3 * P3 + 4 * col);
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (int64_t g = (int64_t)blockIdx.x * rpb + r0; g < groups; g += (int64_t)gridDim.x * rpb) {
        const float4 go = *reinterpret_cast<const float4*>(gout + g * c3 + 4 * col);
        const float4 z = *reinterpret_cast<const float4*>(zmax + g * c3 + 4 * col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (f4(z, e) - f4(mu, e)) * f4(is, e);
            const float dyh = (xh * f4(ga, e) + f4(be, e) > 0.f) ? f4(go, e) : 0.f;
            a[e] += dyh;
            b[e] += (double)dyh * xh;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = a[e]; red[1][threadIdx.x][e] = b[e]; }
    __syncthreads();
    double* part = partial + (size_t)blockIdx.x * SS_PART;
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, ch = threadIdx.x & 63;
        double s = 0.0;
        if (ch < c3)
            for (int r = 0; r < rpb; ++r) s += red[which][r * cg + ch / 4][ch & 3];
        part[threadIdx.x] = s;
    }
}

// STAGE 3: down from the pooled gradient to dz2'; STAGE 2: from dz2' to dz1'; STAGE 1: from dz1' to dW1.
template <int R3, int NS, int STAGE, int W>
__global__ __launch_bounds__(W * 64) void ss_bwd_kernel(const SsParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = wave_id(), lane = lane_id(), h = lane >> 5, j = lane & 31;
    float* lds_all = smem + SS_LDS_WPACK + SS_LDS_STATE;
    float* lds = lds_all + w * SS_LDS_WAVE;
    ss_stage_constants(p, smem, R3);
    float* zt = lds;                      // [<= 64][33]: the gradient block(s), transposed
    float* yt = lds + 64 * SS_LD;         // [32][33]: the activation block, transposed
    SsFrag<4, 1> w1;
    SsFrag<16, 1> w2;
    SsFrag<16, R3> w3;
    SsFrag<16, 1> w2t;
    SsFrag<16 * R3, 1> w3t;
    w1.load(smem, lane);
    w2.load(smem + ss_off_w2(), lane);
    w2t.load(smem + ss_off_w2t(R3), lane);
    w3.load(smem + ss_off_w3(), lane);
    w3t.load(smem + ss_off_w3t(R3), lane);
    const float* st1 = smem + SS_LDS_WPACK + ss_state_base(1);
    const float* st2 = smem + SS_LDS_WPACK + ss_state_base(2);
    const float* st3 = smem + SS_LDS_WPACK + ss_state_base(3);
    constexpr int NBW = STAGE == 3 ? R3 : 1;          // row blocks of this pass's weight gradient
    f32x16 dw[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) dw[b][i] = 0.f;
    float s1[1][16], s2[1][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1[0][i] = 0.f; s2[0][i] = 0.f; }
    if (STAGE == 1) {                                  // x0^T: rows 0..7 are rewritten per tile, rows 8..31 stay zero
        for (int e = lane; e < 32 * SS_LD; e += 64) yt[e] = 0.f;
        ss_wave_fence();
    }
    const int cdn = STAGE == 3 ? p.c2 : p.c1;         // channels of the gradient tensor this pass writes (STAGE >= 2)
    const int cup = STAGE == 2 ? p.c2 : p.c1;         // ... and of the one it reads (STAGE <= 2)

    SsWalk walk;
    walk.start(p, (int64_t)blockIdx.x * W + w, (int64_t)gridDim.x * W, j, h);
    for (; walk.tile < p.tiles; walk.tile += walk.stride) {
        const int64_t tile = walk.tile;
        const int64_t tok = tile * 32 + j;
        asm volatile("" ::: "memory");     // weights and per-channel constants are re-read from LDS per tile, not hoisted into registers
        float x0[4];
        walk.fetch(p, j, h, x0);
        // the gradient this pass starts from, requested before the recompute chain
        const int64_t g = NS == 32 ? tile : tile * 2 + (j >> 4);
        float4 gin[1][4];
        if constexpr (STAGE == 3) {
            // (loaded where they are used: with two waves per SIMD the other wave covers the latency, and 40 registers matter)
        } else {
            const float* row = p.dz_in + tok * cup;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                gin[0][q] = 8 * q + 4 * h < cup ? *reinterpret_cast<const float4*>(row + 8 * q + 4 * h) : make_float4(0, 0, 0, 0);
        }
        f32x16 a1[1], xh1, a2[1], xh2, d[1];
        ss_mm<4, 1>(a1, SsX0{x0}, w1);
        ss_bn_relu<true>(a1[0], xh1, st1, 32, 0, h);                  // a1 = y1
        if (STAGE >= 2) {
            ss_mm<16, 1>(a2, SsAcc<1>{a1}, w2);
            ss_bn_relu<true>(a2[0], xh2, st2, 32, 0, h);              // a2 = y2
        }
        if constexpr (STAGE == 3) {
            // one 32-channel block of layer 3 at a time (registers: the pass fits two waves per SIMD this way); the order of
            // every sum is that of the whole-width form
            const int slot = j & (NS - 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) d[0][i] = 0.f;
            ss_to_lds(yt, a2[0], 0, lane);
#pragma unroll
            for (int rb = 0; rb < R3; ++rb) {
                f32x16 zb[1];
                ss_mm_block<16, R3>(zb[0], SsAcc<1>{a2}, w3, rb);
                // dz3 = gamma3 invstd3 (dyh - mean(dyh) - xh3 mean(dyh xh3)), dyh = the pooled gradient at the arg-max slot
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const SsChan c = ss_chan(st3, 32 * R3, rb, q, h);
                    const float4 m1 = *reinterpret_cast<const float4*>(st3 + 4 * 32 * R3 + rb * 32 + 8 * q + 4 * h);
                    const float4 m2 = *reinterpret_cast<const float4*>(st3 + 5 * 32 * R3 + rb * 32 + 8 * q + 4 * h);
                    const float4 go = *reinterpret_cast<const float4*>(p.gout + g * p.c3 + rb * 32 + 8 * q + 4 * h);
                    const uchar4 ar = *reinterpret_cast<const uchar4*>(p.arg_in + g * p.c3 + rb * 32 + 8 * q + 4 * h);
                    const int av[4] = {ar.x, ar.y, ar.z, ar.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (zb[0][4 * q + e] - f4(c.mu, e)) * f4(c.is, e);
                        const float dyh = (av[e] == slot && xh * f4(c.ga, e) + f4(c.be, e) > 0.f) ? f4(go, e) : 0.f;
                        zb[0][4 * q + e] = f4(c.ga, e) * f4(c.is, e) * (dyh - (f4(m1, e) + xh * f4(m2, e)));
                    }
                }
                ss_to_lds(zt, zb[0], 0, lane);
                ss_wave_fence();
                f32x16 dwb[1] = {dw[rb]};
                ss_dw_tile<1>(dwb, zt, yt, lane);                      // dW3[block rb] += dz3 y2^T
                dw[rb] = dwb[0];
                ss_mm_slice<16 * R3>(d[0], SsAcc<1>{zb}, w3t, rb);     // dy2 += W3^T[:, block rb] dz3
                ss_wave_fence();                                       // zt is rewritten by the next block
            }
        } else {
            // the gradient this pass starts from: dz' (tokens, cup), already masked by its ReLU
            f32x16 dzp;
            const float* stl = STAGE == 2 ? st2 : st1;
            const f32x16& xh = STAGE == 2 ? xh2 : xh1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = gin[0][q];
                const SsChan c = ss_chan(stl, 32, 0, q, h);
                const float4 m1 = *reinterpret_cast<const float4*>(stl + 4 * 32 + 8 * q + 4 * h);
                const float4 m2 = *reinterpret_cast<const float4*>(stl + 5 * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dzp[4 * q + e] = f4(c.ga, e) * f4(c.is, e) * (f4(v, e) - (f4(m1, e) + xh[4 * q + e] * f4(m2, e)));
            }
            ss_to_lds(zt, dzp, 0, lane);
            if (STAGE == 2) {
                ss_to_lds(yt, a1[0], 0, lane);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) yt[(2 * t + h) * SS_LD + j] = x0[t];      // x0 channel k = 2 t + h (k < 8)
            }
            ss_wave_fence();
            ss_dw_tile<1>(dw, zt, yt, lane);                           // dW2 += dz2 y1^T  |  dW1 += dz1 x0^T
            if constexpr (STAGE == 2) {
                f32x16 dz2[1] = {dzp};
                ss_mm<16, 1>(d, SsAcc<1>{dz2}, w2t);                   // dy1 = W2^T dz2
            }
        }
        ss_wave_fence();
        if (STAGE >= 2) {
            // dz' = dy [y > 0] one layer down: to HBM, and its BatchNorm-backward sums
            const f32x16& y = STAGE == 3 ? a2[0] : a1[0];
            const f32x16& xh = STAGE == 3 ? xh2 : xh1;
            float* row = p.dz_out + tok * cdn;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = y[4 * q + e] > 0.f ? d[0][4 * q + e] : 0.f;
                    s1[0][4 * q + e] += o[e];
                    s2[0][4 * q + e] += o[e] * xh[4 * q + e];
                }
                if (8 * q + 4 * h < cdn) *reinterpret_cast<float4*>(row + 8 * q + 4 * h) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    if (STAGE >= 2) ss_store_partial<1, W>(s1, s2, p.partial + (size_t)blockIdx.x * SS_PART, lds_all);
    ss_store_dw<NBW, W>(dw, p.dw_partial + (size_t)blockIdx.x * SS_DW, lds_all);
}

// ---- finalize kernels (one workgroup of 1024 threads = 64 channels x 16 slices of the partials) -----------------------
__device__ __forceinline__ bool ss_sum_partials(const double* __restrict__ partial, int nblocks, double& a, double& b, int& ch) {
    __shared__ double red[2][16][64];
    ch = threadIdx.x & 63;
    const int part = threadIdx.x >> 6;
    double x = 0.0, y = 0.0;
    for (int k = part; k < nblocks; k += 16) { x += partial[(size_t)k * SS_PART + ch]; y += partial[(size_t)k * SS_PART + 64 + ch]; }
    red[0][part][ch] = x; red[1][part][ch] = y;
    __syncthreads();
    if (part != 0) return false;
    a = 0.0; b = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { a += red[0][q][ch]; b += red[1][q][ch]; }
    return true;
}

// forward: mean / invstd (biased variance) + padded gamma / beta into the state, running statistics (unbiased, momentum)
__global__ __launch_bounds__(1024) void ss_finalize_fwd_kernel(const double* __restrict__ partial, int nblocks, int c, int P,
                                                                int64_t rows, float eps, float momentum,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ st, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var) {
    double s1, s2;
    int ch;
    if (!ss_sum_partials(partial, nblocks, s1, s2, ch) || ch >= P) return;
    if (ch >= c) {      // padded channel: z == 0 there; gamma = beta = 0 keeps y == 0
        st[ch] = 0.f; st[P + ch] = 0.f; st[2 * P + ch] = 0.f; st[3 * P + ch] = 0.f;
        return;
    }
    const double mean = s1 / (double)rows;
    double var = s2 / (double)rows - mean * mean;
    var = var < 0 ? 0 : var;
    st[ch] = (float)mean;
    st[P + ch] = (float)(1.0 / sqrt(var + (double)eps));
    st[2 * P + ch] = gamma[ch];
    st[3 * P + ch] = beta[ch];
    if (running_mean) {
        const double unbiased = rows > 1 ? var * ((double)rows / (double)(rows - 1)) : var;
        running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
        running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
    }
}

// backward: block 0: dbeta = sum dyh, dgamma = sum dyh xh, their means into the state (m1, m2); blocks 1..: the weight
// gradient of the pass in front: dW[row][col] = sum over the workgroup partials, fixed order
__global__ __launch_bounds__(1024) void ss_finalize_bwd_kernel(const double* __restrict__ partial, int nblocks, int c, int P, int64_t rows,
                                                                float* __restrict__ st, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                const float* __restrict__ dw_partial, int dw_blocks, int dw_rows, int dw_cols,
                                                                float* __restrict__ dw) {
    if (blockIdx.x == 0) {
        if (partial == nullptr) return;
        double s1, s2;
        int ch;
        if (!ss_sum_partials(partial, nblocks, s1, s2, ch) || ch >= P) return;
        st[4 * P + ch] = ch < c ? (float)(s1 / (double)rows) : 0.f;
        st[5 * P + ch] = ch < c ? (float)(s2 / (double)rows) : 0.f;
        if (ch < c) { dbeta[ch] = (float)s1; dgamma[ch] = (float)s2; }
        return;
    }
    // 64 elements x 16 slices of the workgroup partials per block; slices summed in fixed order
    __shared__ double red[16][64];
    const int el = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int e = (blockIdx.x - 1) * 64 + el;
    const bool ok = dw != nullptr && e < dw_rows * dw_cols;
    const int row = ok ? e / dw_cols : 0, col = ok ? e % dw_cols : 0;
    double a = 0.0;
    if (ok)
        for (int k = part; k < dw_blocks; k += 16) a += (double)dw_partial[(size_t)k * SS_DW + row * 32 + col];
    red[part][el] = a;
    __syncthreads();
    if (part == 0 && ok) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][el];
        dw[e] = (float)t;
    }
}

template <typename K>
static bool ss_lds_ok(K kern, PerDevice<bool>& once, int bytes) {
    return once.get([kern, bytes] { return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess; });
}
static int ss_blocks_for(int64_t tiles, int waves) {      // one workgroup per CU (LDS), every wave walks >= 1 tile
    const int64_t want = divup64(tiles, waves);
    return (int)(want < 256 ? want : 256);
}

template <int R3, int NS, int STAGE>
static bool ss_launch_fwd_stage(const SsParams& p, hipStream_t s, int& blocks) {
    constexpr int W = SS_WAVES;
    static PerDevice<bool> once;
    constexpr int bytes = ss_lds_floats(W) * (int)sizeof(float);
    if (!ss_lds_ok(ss_fwd_kernel<R3, NS, STAGE, W>, once, bytes)) return false;
    blocks = ss_blocks_for(p.tiles, W);
    hipLaunchKernelGGL((ss_fwd_kernel<R3, NS, STAGE, W>), dim3(blocks), dim3(W * 64), bytes, s, p);
    return true;
}
template <int R3, int NS>
static bool ss_launch_fwd(int stage, const SsParams& p, hipStream_t s, int& blocks) {
    switch (stage) {
        case 1: return ss_launch_fwd_stage<R3, NS, 1>(p, s, blocks);
        case 2: return ss_launch_fwd_stage<R3, NS, 2>(p, s, blocks);
        case 3: return ss_launch_fwd_stage<R3, NS, 3>(p, s, blocks);
        default: return ss_launch_fwd_stage<R3, NS, 4>(p, s, blocks);
    }
}
template <int R3, int NS, int STAGE>
static bool ss_launch_bwd_stage(const SsParams& p, hipStream_t s, int& blocks) {
    constexpr int W = SS_WAVES;
    static PerDevice<bool> once;
    constexpr int bytes = ss_lds_floats(W) * (int)sizeof(float);
    if (!ss_lds_ok(ss_bwd_kernel<R3, NS, STAGE, W>, once, bytes)) return false;
    blocks = ss_blocks_for(p.tiles, W);
    hipLaunchKernelGGL((ss_bwd_kernel<R3, NS, STAGE, W>), dim3(blocks), dim3(W * 64), bytes, s, p);
    return true;
}
template <int R3, int NS>
static bool ss_launch_bwd(int stage, const SsParams& p, hipStream_t s, int& blocks) {
    switch (stage) {
        case 3: return ss_launch_bwd_stage<R3, NS, 3>(p, s, blocks);
        case 2: return ss_launch_bwd_stage<R3, NS, 2>(p, s, blocks);
        default: return ss_launch_bwd_stage<R3, NS, 1>(p, s, blocks);
    }
}

static int ss_check_shape(const char* what, int b, int n, int m, int c, int ns, int c1, int c2, int c3) {
    PDA_REQUIRE(b >= 1 && n >= 1 && m >= 1, "%s: bad size (b=%d n=%d m=%d)", what, b, n, m);
    const bool ok = c >= 0 && c <= 5 && (ns == 16 || ns == 32) && (c1 == 16 || c1 == 32) && (c2 == 16 || c2 == 32) &&
                    (c3 == 32 || c3 == 64) && (32 / ns) * c3 == 64 && ((int64_t)b * m * ns) % 32 == 0;
    if (!ok) {
        set_error("%s: no kernel built for the chain %d -> %d -> %d -> %d over %d neighbours (built: 3 + c <= 8 inputs, "
                  "widths 16 | 32, 16 | 32, and (nsample, c3) = (16, 32) | (32, 64))", what, 3 + c, c1, c2, c3, ns);
        return PDA_ERR_UNSUPPORTED;
    }
    return PDA_OK;
}

// workspace layout (bytes, all 256-byte aligned): state | wpack | partial | dw_partial
constexpr size_t SS_WS_STATE = 0;
constexpr size_t SS_WS_WPACK = 4096;
constexpr size_t SS_WS_PARTIAL = SS_WS_WPACK + 32768;
constexpr size_t SS_WS_DW = SS_WS_PARTIAL + (size_t)SS_MAX_BLOCKS * SS_PART * sizeof(double);
constexpr size_t SS_WS_BYTES = SS_WS_DW + (size_t)SS_MAX_BLOCKS * SS_DW * sizeof(float);

}  // namespace pda

PDA_API int64_t pda_sa_small_train_workspace_bytes(void) { return (int64_t)pda::SS_WS_BYTES; }

PDA_API int pda_sa_small_train_supported(int c, int ns, int c1, int c2, int c3, int64_t tokens) {
    return c >= 0 && c <= 5 && (ns == 16 || ns == 32) && (c1 == 16 || c1 == 32) && (c2 == 16 || c2 == 32) && (c3 == 32 || c3 == 64) &&
           (32 / ns) * c3 == 64 && tokens > 0 && tokens % 32 == 0;
}

PDA_API int pda_sa_small_train_fwd(const float* xyz, const float* new_xyz, const float* feat_pm, const int32_t* idx, const float* w1,
                                   const float* w2, const float* w3, const float* const* gamma, const float* const* beta,
                                   float* const* running_mean, float* const* running_var, const float* eps, const float* momentum,
                                   void* workspace, float* out, float* zmax, uint8_t* arg, int b, int n, int m, int c, int ns, int c1,
                                   int c2, int c3, pda_stream_t stream) {
    using namespace pda;
    const int rc = ss_check_shape("pda_sa_small_train_fwd", b, n, m, c, ns, c1, c2, c3);
    if (rc != PDA_OK) return rc;
    PDA_REQUIRE(xyz && new_xyz && idx && w1 && w2 && w3 && gamma && beta && running_mean && running_var && eps && momentum && workspace &&
                out && zmax && arg && (feat_pm || c == 0), "pda_sa_small_train_fwd: null pointer");
    PDA_REQUIRE(((uintptr_t)workspace & 255) == 0, "pda_sa_small_train_fwd: workspace must be 256-byte aligned");
    const hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* state = (float*)(ws + SS_WS_STATE);
    float* wpack = (float*)(ws + SS_WS_WPACK);
    const int R3 = c3 / 32;
    hipLaunchKernelGGL(ss_pack_kernel, dim3(8), dim3(256), 0, s, w1, w2, w3, wpack, 3 + c, c1, c2, c3, R3);
    SsParams p{};
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat = feat_pm; p.idx = idx; p.wpack = wpack; p.state = state;
    p.partial = (double*)(ws + SS_WS_PARTIAL); p.out = out; p.zmax = zmax; p.arg = arg;
    p.n = n; p.m = m; p.c = c; p.ns = ns; p.c1 = c1; p.c2 = c2; p.c3 = c3;
    const int64_t tokens = (int64_t)b * m * ns;
    p.tiles = tokens / 32;
    int blocks = 0;
    const int cs[3] = {c1, c2, c3}, Ps[3] = {32, 32, 32 * R3};
    for (int stage = 1; stage <= 4; ++stage) {
        const bool ok = R3 == 1 ? ss_launch_fwd<1, 16>(stage, p, s, blocks) : ss_launch_fwd<2, 32>(stage, p, s, blocks);
        PDA_REQUIRE(ok, "pda_sa_small_train_fwd: dynamic LDS refused");
        if (stage <= 3)
            hipLaunchKernelGGL(ss_finalize_fwd_kernel, dim3(1), dim3(1024), 0, s, (const double*)p.partial, blocks, cs[stage - 1],
                               Ps[stage - 1], tokens, eps[stage - 1], momentum[stage - 1], gamma[stage - 1], beta[stage - 1],
                               state + ss_state_base(stage), running_mean[stage - 1], running_var[stage - 1]);
    }
    return check_launch("pda_sa_small_train_fwd");
}

PDA_API int pda_sa_small_train_bwd(const float* xyz, const float* new_xyz, const float* feat_pm, const int32_t* idx, const float* grad_out,
                                   const float* zmax, const uint8_t* arg, void* workspace, float* dz2, float* dz1, float* dw1, float* dw2,
                                   float* dw3, float* const* dgamma, float* const* dbeta, int b, int n, int m, int c, int ns, int c1,
                                   int c2, int c3, pda_stream_t stream) {
    using namespace pda;
    const int rc = ss_check_shape("pda_sa_small_train_bwd", b, n, m, c, ns, c1, c2, c3);
    if (rc != PDA_OK) return rc;
    PDA_REQUIRE(xyz && new_xyz && idx && grad_out && zmax && arg && workspace && dz2 && dz1 && dw1 && dw2 && dw3 && dgamma && dbeta &&
                (feat_pm || c == 0), "pda_sa_small_train_bwd: null pointer");
    const hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* state = (float*)(ws + SS_WS_STATE);
    SsParams p{};
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat = feat_pm; p.idx = idx; p.wpack = (const float*)(ws + SS_WS_WPACK); p.state = state;
    p.partial = (double*)(ws + SS_WS_PARTIAL); p.dw_partial = (float*)(ws + SS_WS_DW);
    p.gout = grad_out; p.arg_in = arg;
    p.n = n; p.m = m; p.c = c; p.ns = ns; p.c1 = c1; p.c2 = c2; p.c3 = c3;
    const int64_t tokens = (int64_t)b * m * ns, groups = (int64_t)b * m;
    p.tiles = tokens / 32;
    int blocks = 0;
    const int R3 = c3 / 32;
    // E0: BatchNorm-3 backward sums from the pooled side
    const int pb = (int)(divup64(groups, 256 / (c3 / 4)) < 256 ? divup64(groups, 256 / (c3 / 4)) : 256);
    hipLaunchKernelGGL(ss_pool_sums_kernel, dim3(pb), dim3(256), 0, s, grad_out, zmax, (const float*)(state + ss_state_base(3)), 32 * R3, c3,
                       groups, p.partial);
    hipLaunchKernelGGL(ss_finalize_bwd_kernel, dim3(1), dim3(1024), 0, s, (const double*)p.partial, pb, c3, 32 * R3, tokens,
                       state + ss_state_base(3), dgamma[2], dbeta[2], (const float*)nullptr, 0, 0, 0, (float*)nullptr);
    // B3: dW3, dz2', BatchNorm-2 sums
    p.dz_out = dz2;
    { const bool ok = R3 == 1 ? ss_launch_bwd<1, 16>(3, p, s, blocks) : ss_launch_bwd<2, 32>(3, p, s, blocks);
      PDA_REQUIRE(ok, "pda_sa_small_train_bwd: dynamic LDS refused"); }
    hipLaunchKernelGGL(ss_finalize_bwd_kernel, dim3(1 + divup(c3 * c2, 64)), dim3(1024), 0, s, (const double*)p.partial, blocks, c2, 32, tokens,
                       state + ss_state_base(2), dgamma[1], dbeta[1], (const float*)p.dw_partial, blocks, c3, c2, dw3);
    // B2: dW2, dz1', BatchNorm-1 sums
    p.dz_in = dz2; p.dz_out = dz1;
    { const bool ok = R3 == 1 ? ss_launch_bwd<1, 16>(2, p, s, blocks) : ss_launch_bwd<2, 32>(2, p, s, blocks);
      PDA_REQUIRE(ok, "pda_sa_small_train_bwd: dynamic LDS refused"); }
    hipLaunchKernelGGL(ss_finalize_bwd_kernel, dim3(1 + divup(c2 * c1, 64)), dim3(1024), 0, s, (const double*)p.partial, blocks, c1, 32, tokens,
                       state + ss_state_base(1), dgamma[0], dbeta[0], (const float*)p.dw_partial, blocks, c2, c1, dw2);
    // B1: dW1
    p.dz_in = dz1; p.dz_out = nullptr;
    { const bool ok = R3 == 1 ? ss_launch_bwd<1, 16>(1, p, s, blocks) : ss_launch_bwd<2, 32>(1, p, s, blocks);
      PDA_REQUIRE(ok, "pda_sa_small_train_bwd: dynamic LDS refused"); }
    hipLaunchKernelGGL(ss_finalize_bwd_kernel, dim3(1 + divup(c1 * (3 + c), 64)), dim3(1024), 0, s, (const double*)nullptr, 0, 0, 0, tokens,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)p.dw_partial, blocks, c1, 3 + c, dw1);
    return check_launch("pda_sa_small_train_bwd");
}
