// fps.hip -- furthest point sampling for gfx950, index-exact w.r.t. the reference kernels
// farthest_point_sampling_kernel / furthest_point_sampling_with_dist_kernel
// (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:93-209, :256-371).
//
// The algorithm is m-1 strictly dependent iterations, each an arg-max over N points, so it is
// latency-bound: what matters is the length of one iteration, not bandwidth.
//
// fps_reg_kernel<P> (N <= 1024*P <= 24576: every BASELINE 16384-pt shape)
//   * one 1024-lane workgroup per scene; lane t OWNS points t, t+1024, ... exactly like thread
//     t of the reference's block<1024>, but keeps their x,y,z and running min-distance `temp`
//     in REGISTERS for the whole kernel: per iteration no xyz/temp byte moves through LDS, L2
//     or HBM (the reference re-reads 16 B and writes 4 B per point per iteration);
//   * the 10-barrier shared-memory tree (:143-203) becomes: DPP wave reduction -> one 8-byte
//     LDS slot per wave -> ONE barrier -> every wave reduces the 16 slots itself; LDS slots are
//     double-buffered so no second barrier is needed;
//   * the tree's tie-breaking is reproduced exactly by reducing with the total order
//     (max min-dist, then min (bitreverse(k mod bs), k / bs)), bs = the reference's block size
//     (pinned by tests/test_oracle_known_answers.py);
//   * the winner's coordinates are fetched with a scalar load (wave-uniform address).
//   Measured dead ends (MI355X, 16384->4096, B=2; profiles/r01_fps_variants.txt): packing two
//   points per v_pk_{add,mul,fma}_f32 is 13 % SLOWER (5.40 vs 4.79 ms: packed f32 issues at half
//   rate and costs extra moves), 512 lanes x 32 points/lane 22-36 % slower (fewer waves to
//   cover the reduction latency).
// fps_stream_kernel<WITH_DIST> (any N; also the (B,N,N) distance-matrix variant)
//   * same reduction, but xyz (or the dist row) and temp stream from L2/HBM each iteration.
// Algorithmic bytes: (m-1)*N*20 + m*4 per scene; compulsory bytes N*16 + m*4.
#include "pda_common.h"

namespace pda {

constexpr int FPS_THREADS = 1024;
constexpr int FPS_WAVES = FPS_THREADS / PDA_WAVE;

// tie-break value of point k under the reference's tree with block size 2^L:
// smaller = preferred.  High bits: bit-reversed lane (k mod bs); low bits: k / bs.
__device__ __forceinline__ uint32_t fps_tiebreak(uint32_t k, int L) {
    const uint32_t lane_bits = k & ((1u << L) - 1u);
    return __builtin_bitreverse32(lane_bits) | (k >> L);  // L == 0: bitreverse(0) | k
}
__device__ __forceinline__ uint32_t fps_tiebreak_decode(uint32_t T, int L) {
    if (L == 0) return T;
    const uint32_t lowmask = (1u << (32 - L)) - 1u;
    return ((T & lowmask) << L) | __builtin_bitreverse32(T & ~lowmask);
}

// Workgroup arg-max.  Each lane contributes (best, T); returns the winning point index,
// identical in every lane (wave-uniform).  `slots` is this iteration's LDS buffer.
__device__ __forceinline__ int fps_block_argmax(float best, uint32_t T, uint2* slots, int nwaves,
                                                int w, int lane, int L) {
    const float wmax = wave_max_f32(best);
    const uint32_t wT = wave_min_u32(best == wmax ? T : 0xffffffffu);
    if (lane == 0) slots[w] = make_uint2(__builtin_bit_cast(uint32_t, wmax), wT);
    __syncthreads();
    uint2 s = make_uint2(__builtin_bit_cast(uint32_t, -1.0f), 0xffffffffu);
    if (lane < nwaves) s = slots[lane];
    const float v = __builtin_bit_cast(float, s.x);
    const float bmax = row0_max_f32(v);  // nwaves <= 16: the slots sit in lanes 0..15
    const uint32_t bT = row0_min_u32(v == bmax ? s.y : 0xffffffffu);
    return (int)fps_tiebreak_decode(bT, L);
}

template <int P>
__global__ __launch_bounds__(FPS_THREADS) void fps_reg_kernel(const float* __restrict__ xyz_all,
                                                               float* __restrict__ temp_all,
                                                               int32_t* __restrict__ idx_all, int n,
                                                               int m, int L) {
    __shared__ uint2 slots[2][FPS_WAVES];
    const int t = threadIdx.x;
    const int lane = lane_id();
    const int w = wave_id();
    const int nwaves = (int)(blockDim.x >> 6);
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    float* __restrict__ temp = temp_all + (size_t)blockIdx.x * n;
    int32_t* __restrict__ idx = idx_all + (size_t)blockIdx.x * m;

    float px[P], py[P], pz[P], tp[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = t + i * FPS_THREADS;
        if (k < n) {
            px[i] = xyz[k * 3 + 0]; py[i] = xyz[k * 3 + 1]; pz[i] = xyz[k * 3 + 2];
            tp[i] = temp[k];
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            tp[i] = -1.f;  // min(d, -1) = -1 never beats best = -1: slot is inert
        }
    }

    int old = 0;
    if (t == 0) idx[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
        float best = -1.f;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const float d = sqdist3(px[i], py[i], pz[i], x1, y1, z1);  // (x2 - x1), :133
            const float d2 = fminf(d, tp[i]);
            tp[i] = d2;
            const bool g = d2 > best;  // strict: lowest k wins inside a lane (:136-137)
            bi = g ? i : bi;
            best = g ? d2 : best;
        }
        const uint32_t T = fps_tiebreak((uint32_t)(t + bi * FPS_THREADS), L);
        old = fps_block_argmax(best, T, slots[j & 1], nwaves, w, lane, L);
        if (t == 0) idx[j] = old;
    }

#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = t + i * FPS_THREADS;
        if (k < n) temp[k] = tp[i];
    }
}

template <bool WITH_DIST>
__global__ __launch_bounds__(FPS_THREADS) void fps_stream_kernel(const float* __restrict__ data_all,
                                                                  float* __restrict__ temp_all,
                                                                  int32_t* __restrict__ idx_all, int n,
                                                                  int m, int L) {
    __shared__ uint2 slots[2][FPS_WAVES];
    const int t = threadIdx.x;
    const int lane = lane_id();
    const int w = wave_id();
    const int nwaves = (int)(blockDim.x >> 6);
    const float* __restrict__ data = data_all + (size_t)blockIdx.x * n * (WITH_DIST ? (size_t)n : 3);
    float* __restrict__ temp = temp_all + (size_t)blockIdx.x * n;
    int32_t* __restrict__ idx = idx_all + (size_t)blockIdx.x * m;

    int old = 0;
    if (t == 0) idx[0] = 0;
    for (int j = 1; j < m; ++j) {
        float x1 = 0.f, y1 = 0.f, z1 = 0.f;
        if (!WITH_DIST) { x1 = data[old * 3 + 0]; y1 = data[old * 3 + 1]; z1 = data[old * 3 + 2]; }
        const float* __restrict__ drow = data + (size_t)old * n;  // WITH_DIST: row `old` (:294)
        float best = -1.f;
        int bk = 0;
        for (int k = t; k < n; k += FPS_THREADS) {
            float d;
            if (WITH_DIST) d = drow[k];
            else d = sqdist3(data[k * 3 + 0], data[k * 3 + 1], data[k * 3 + 2], x1, y1, z1);
            const float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            const bool g = d2 > best;
            bk = g ? k : bk;
            best = g ? d2 : best;
        }
        const uint32_t T = best >= 0.f ? fps_tiebreak((uint32_t)bk, L) : 0xffffffffu;
        old = fps_block_argmax(best, T, slots[j & 1], nwaves, w, lane, L);
        if (t == 0) idx[j] = old;
    }
}

static int ilog2(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}

static int launch_fps(bool with_dist, const float* data, float* temp, int32_t* idx, int b, int n, int m,
                      hipStream_t stream, const char* what) {
    PDA_REQUIRE(b >= 0 && n >= 0, "%s: negative size (b=%d n=%d)", what, b, n);
    if (b == 0 || m <= 0) return PDA_OK;  // m <= 0: the reference kernel returns at once (:101)
    PDA_REQUIRE(n >= 1, "%s: n == 0 but m == %d", what, m);
    PDA_REQUIRE(data && temp && idx, "%s: null pointer", what);
    PDA_REQUIRE((int64_t)n * 3 < INT32_MAX, "%s: n too large", what);
    const int L = ilog2(pda_opt_n_threads(n));  // the reference's block size fixes the tie-break
    const int threads = n >= FPS_THREADS ? FPS_THREADS : divup(n, PDA_WAVE) * PDA_WAVE;
    dim3 grid(b), block(threads);
    if (with_dist) {
        hipLaunchKernelGGL(fps_stream_kernel<true>, grid, block, 0, stream, data, temp, idx, n, m, L);
        return check_launch(what);
    }
    const int P = divup(n, FPS_THREADS);
#define PDA_FPS_CASE(PP)                                                                          \
    hipLaunchKernelGGL(fps_reg_kernel<PP>, grid, block, 0, stream, data, temp, idx, n, m, L)
    if (P <= 1) PDA_FPS_CASE(1);
    else if (P <= 2) PDA_FPS_CASE(2);
    else if (P <= 4) PDA_FPS_CASE(4);
    else if (P <= 8) PDA_FPS_CASE(8);
    else if (P <= 16) PDA_FPS_CASE(16);
    else if (P <= 24) PDA_FPS_CASE(24);
    else hipLaunchKernelGGL(fps_stream_kernel<false>, grid, block, 0, stream, data, temp, idx, n, m, L);
#undef PDA_FPS_CASE
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_opt_n_threads(int work_size) {
    // cuda_utils.h:10-14: pow_2 = log(double(n)) / log(2.0) truncated; clamp(1 << pow_2, 1, 1024).
    // Computed with the same double expression as the reference's host code.
    if (work_size < 1) return 1;
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

PDA_API int pda_furthest_point_sampling(const float* xyz, float* temp, int32_t* idx, int b, int n, int m,
                                        pda_stream_t stream) {
    return pda::launch_fps(false, xyz, temp, idx, b, n, m, (hipStream_t)stream,
                           "pda_furthest_point_sampling");
}

PDA_API int pda_furthest_point_sampling_with_dist(const float* dist, float* temp, int32_t* idx, int b,
                                                  int n, int m, pda_stream_t stream) {
    return pda::launch_fps(true, dist, temp, idx, b, n, m, (hipStream_t)stream,
                           "pda_furthest_point_sampling_with_dist");
}
