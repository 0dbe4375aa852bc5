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
// fps_chain_kernel<P> (2048 <= N <= 16384: several samples per synchronisation, see the comment at the kernel)
// fps_chain_coop_kernel<16> (16384 < N <= 65536: K = ceil(N/16384) workgroups per scene, several samples per exchange)
// fps_pruned_kernel<P,COOP> (their one-sample-per-round forms: PDA_FPS_NO_CHAIN=1 / PDA_FPS_COOP_CHAIN=0)
// fps_stream_kernel<WITH_DIST> (any other N; also the (B,N,N) distance-matrix variant)
//   * same reduction, but xyz (or the dist row) and temp stream from L2/HBM each iteration.
// Algorithmic bytes: (m-1)*N*20 + m*4 per scene; compulsory bytes N*16 + m*4.
#include "pda_common.h"

#include <stdlib.h>

#include <algorithm>
#include <atomic>

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
    lds_barrier();  // not __syncthreads(): the idx[j] store of the previous round stays in flight
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

// ---------------------------------------------------------------------------------------------
// fps_pruned_kernel<P>: exact FPS with spatial pruning (2048 <= N <= 16384).
//
// The update  temp[k] = min(temp[k], |p_k - s|^2)  changes nothing for points farther from the new
// sample s than their current temp.  Points arrive shuffled (data_processor.py:93-103), so the
// kernel first sorts them along a space-filling curve IN LDS (18-bit adaptive Morton key | 14-bit
// index, bitonic sort) and gives every lane 16 CONSECUTIVE sorted points: a lane owns a compact
// cluster (box of a few metres), a wave a compact region.  Per round each lane tests the new
// sample against its cluster's bounding box: when  0.9999 * dist2(box, s) >= max temp of the lane
// for ALL 64 lanes, the wave skips the 16-point scan and re-publishes its cached candidate.
// The factor 0.9999 (>> the ~1e-6 relative rounding of the distance expressions) makes the test
// conservative, so skipped updates are exactly the no-op updates: indices and the final `temp`
// stay bit-identical to the reference.  Ties are resolved with the same total order as
// fps_reg_kernel: inside a lane the slots are ordered by the tie-break value T(k) (sorting
// network at set-up) and scanned with a strict '>', across lanes/waves by (max, min T).
// The winner's coordinates travel through the LDS slots (no dependent global load per round).
// Measured (MI355X, 16384->4096, B=2, profiles/r01_fps_variants.txt): 1.85 of 16 waves scan per
// round on average; 4.74 ms (fps_reg_kernel) -> 3.74 ms.  The round is now bounded by the one
// scanning wave (~250 dependent-ish instructions alone on its SIMD, 0.60 us) plus two barriers
// and three LDS round trips (0.31 us).  PDA_FPS_NO_PRUNE=1 selects the unpruned kernel for A/B.
// readlane(v[js], lstar) for a wave-uniform, run-time slot js: registers cannot be indexed at run
// time, so descend a binary tree of wave-uniform branches (log2 P levels) to the static slot.
template <int LO, int N, int P>
__device__ __forceinline__ void fps_extract(int js, int lstar, const float (&px)[P], const float (&py)[P],
                                            const float (&pz)[P], const uint32_t (&kT)[P], float& cx, float& cy,
                                            float& cz, uint32_t& wT) {
    if constexpr (N == 1) {
        cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px[LO]), lstar));
        cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py[LO]), lstar));
        cz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz[LO]), lstar));
        wT = (uint32_t)__builtin_amdgcn_readlane((int)kT[LO], lstar);
    } else {
        if (js < LO + N / 2) fps_extract<LO, N / 2, P>(js, lstar, px, py, pz, kT, cx, cy, cz, wT);
        else fps_extract<LO + N / 2, N - N / 2, P>(js, lstar, px, py, pz, kT, cx, cy, cz, wT);
    }
}

#ifdef PDA_FPS_STATS
__device__ unsigned long long g_fps_stats[16];  // [0] wave-scans, [1] wave-rounds, [2] hit lanes
#endif

// Cooperative form (COOP): K workgroups share one scene (16384 < N <= 65536).  Workgroup g owns the
// contiguous index range [g*ceil(N/K), ...) -- an arbitrary quarter of the shuffled cloud, sorted and
// pruned locally exactly as above -- and every round the K local winners are exchanged through
// 8-byte {payload, tag} granules in device memory (relaxed agent-scope 64-bit atomics = sc1
// stores/loads, MI355X_MICROARCH.md "Valid forms": a granule needs no further ordering; the tag is
// (launch epoch, round)).  Each workgroup then reduces the K records with the same total order, so
// all of them continue with the same sample.  Requires the K workgroups of a scene to be resident
// together (host: grid <= 256 workgroups); every poll loop is bounded (FPS_SPIN_LIMIT).
//
// Failure model.  The exchange needs the K workgroups of a scene to be resident together.  The host sizes every
// launch to what the device holds at once (launch_fps: grid <= CUs x 1 workgroup), so a workgroup can only be
// late -- behind another stream's kernel that occupies its CU -- never absent; the polls are bounded all the same
// (g_fps_spin_limit polls of ~0.1 us).  A timeout is neither silent nor fatal: the workgroup records the launch
// epoch in g_fps_fail[region][scene], counts it in g_fps_fail_total and stops; fps_recover_kernel, launched behind
// every cooperative launch on the same stream, finds the mark and recomputes that scene with the streaming
// algorithm (temp re-filled with 1e10, the entry contract), so idx/temp are the exact result either way and no
// unwritten index ever reaches a gather.  pda_fps_coop_timeouts() reports the count to the host.
constexpr int FPS_XBUF_REGIONS = 4, FPS_XBUF_SCENES = 64, FPS_MAX_K = 4, FPS_SPIN_LIMIT = 1 << 20;
__device__ unsigned long long g_fps_xbuf[FPS_XBUF_REGIONS * FPS_XBUF_SCENES * 2 * FPS_MAX_K * 5];
__device__ int g_fps_spin_limit = FPS_SPIN_LIMIT;
__device__ unsigned int g_fps_fail[FPS_XBUF_REGIONS * FPS_XBUF_SCENES];   // launch epoch of a timed-out exchange, per scene
__device__ unsigned long long g_fps_fail_total;                            // workgroups that ever timed out

// Set-up shared by the pruned forms: scene bounding box -> adaptive 18-bit curve key -> bitonic sort in LDS -> lane t takes
// the P consecutive sorted points t*P .. t*P+P-1 (ordered by tie-break value inside the lane) into registers, with the
// lane's bounding box and its running maximum.  skey: NS = 1024 * P words, red: FPS_WAVES * 6 floats (LDS).
template <int P>
__device__ __forceinline__ void fps_sorted_setup(const float* __restrict__ xyz, const float* __restrict__ temp, int n, int k_lo,
                                                 int L, uint32_t* skey, float* red, float (&px)[P], float (&py)[P],
                                                 float (&pz)[P], float (&tp)[P], uint32_t (&kT)[P], int (&kk)[P],
                                                 float (&blo)[3], float (&bhi)[3], float& lbest, int& bi) {
    constexpr int NS = FPS_THREADS * P;
    const int t = threadIdx.x;
    const int lane = lane_id();
    const int w = wave_id();
    const float INF = __builtin_inff();
    // ---- 1. scene bounding box ---------------------------------------------------------
    float lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = t + i * FPS_THREADS;
        if (k < n) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = xyz[k * 3 + a];
                lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float mn = -wave_max_f32(-lo[a]);
        const float mx = wave_max_f32(hi[a]);
        if (lane == 0) { red[w * 6 + a] = mn; red[w * 6 + 3 + a] = mx; }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float mn = INF, mx = -INF;
        for (int q = 0; q < FPS_WAVES; ++q) { mn = fminf(mn, red[q * 6 + a]); mx = fmaxf(mx, red[q * 6 + 3 + a]); }
        lo[a] = mn; hi[a] = mx;
    }
    // ---- 2. adaptive 18-bit key: each level splits the axis whose cells are currently widest
    int bits[3] = {0, 0, 0};
    uint64_t seq = 0;  // 2 bits per level: which axis
    {
        float cell[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        for (int sidx = 0; sidx < 18; ++sidx) {
            const int a = (cell[0] >= cell[1] && cell[0] >= cell[2]) ? 0 : (cell[1] >= cell[2] ? 1 : 2);
            seq |= (uint64_t)a << (2 * sidx);
            if (a == 0) { bits[0]++; cell[0] *= 0.5f; } else if (a == 1) { bits[1]++; cell[1] *= 0.5f; } else { bits[2]++; cell[2] *= 0.5f; }
        }
    }
    float scl[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ext = hi[a] - lo[a];
        scl[a] = ext > 0.f ? (float)(1 << bits[a]) / (ext * 1.0001f) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = t + i * FPS_THREADS;
        uint32_t key = 0xffffffffu;  // padding sorts to the end
        if (k < n) {
            uint32_t q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int qq = (int)((xyz[k * 3 + a] - lo[a]) * scl[a]);
                q[a] = (uint32_t)min(max(qq, 0), (1 << bits[a]) - 1);
            }
            int rem0 = bits[0], rem1 = bits[1], rem2 = bits[2];
            uint32_t code = 0;
            for (int sidx = 0; sidx < 18; ++sidx) {
                const int a = (int)((seq >> (2 * sidx)) & 3);
                uint32_t bit;
                if (a == 0) { rem0--; bit = (q[0] >> rem0) & 1u; }
                else if (a == 1) { rem1--; bit = (q[1] >> rem1) & 1u; }
                else { rem2--; bit = (q[2] >> rem2) & 1u; }
                code = (code << 1) | bit;
            }
            key = (code << 14) | (uint32_t)k;
        }
        skey[k] = key;
    }
    __syncthreads();
    // ---- 3. bitonic sort of skey[0, NS) in LDS -----------------------------------------------
    for (int kk2 = 2; kk2 <= NS; kk2 <<= 1) {
        for (int jj = kk2 >> 1; jj > 0; jj >>= 1) {
            for (int e = t; e < NS / 2; e += FPS_THREADS) {
                const int i0 = ((e & ~(jj - 1)) << 1) | (e & (jj - 1));
                const int i1 = i0 | jj;
                const uint32_t a = skey[i0], b = skey[i1];
                const bool asc = (i0 & kk2) == 0;
                if ((a > b) == asc) { skey[i0] = b; skey[i1] = a; }
            }
            __syncthreads();
        }
    }
    // ---- 4. my P consecutive sorted points, ordered by tie-break value inside the lane ------
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const uint32_t e = skey[t * P + i];
        const bool valid = e != 0xffffffffu;
        kk[i] = valid ? (int)(e & 0x3fffu) : -1;
        kT[i] = valid ? fps_tiebreak((uint32_t)(kk[i] + k_lo), L) : 0xffffffffu;  // global index
    }
#pragma unroll
    for (int k2 = 2; k2 <= P; k2 <<= 1)
#pragma unroll
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1)
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int l2 = i ^ j2;
                if (l2 > i) {
                    const bool asc = (i & k2) == 0;
                    const bool sw = (kT[i] > kT[l2]) == asc;
                    const uint32_t ta = kT[i], tb = kT[l2];
                    const int ka = kk[i], kb = kk[l2];
                    kT[i] = sw ? tb : ta; kT[l2] = sw ? ta : tb;
                    kk[i] = sw ? kb : ka; kk[l2] = sw ? ka : kb;
                }
            }
    blo[0] = blo[1] = blo[2] = INF; bhi[0] = bhi[1] = bhi[2] = -INF;
    lbest = -1.f;
    bi = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        if (kk[i] >= 0) {
            px[i] = xyz[kk[i] * 3 + 0]; py[i] = xyz[kk[i] * 3 + 1]; pz[i] = xyz[kk[i] * 3 + 2];
            tp[i] = temp[kk[i]];
            blo[0] = fminf(blo[0], px[i]); bhi[0] = fmaxf(bhi[0], px[i]);
            blo[1] = fminf(blo[1], py[i]); bhi[1] = fmaxf(bhi[1], py[i]);
            blo[2] = fminf(blo[2], pz[i]); bhi[2] = fmaxf(bhi[2], pz[i]);
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            tp[i] = -1.f;
        }
        const bool g = tp[i] > lbest;
        bi = g ? i : bi;
        lbest = g ? tp[i] : lbest;
    }

}

template <int P, bool COOP>
__global__ __launch_bounds__(FPS_THREADS) void fps_pruned_kernel(const float* __restrict__ xyz_all,
                                                                  float* __restrict__ temp_all,
                                                                  int32_t* __restrict__ idx_all, int n_total,
                                                                  int m, int L, int K, int nb, uint32_t epoch) {
    constexpr int NS = FPS_THREADS * P;  // sort size (power of two)
    __shared__ uint32_t skey[NS];
    __shared__ float red[FPS_WAVES * 6];
    __shared__ uint2 slots[2][FPS_WAVES];
    __shared__ float4 cand[2][FPS_WAVES];
    __shared__ int failflag;
    const int t = threadIdx.x;
    const int lane = lane_id();
    const int w = wave_id();
    const int scene = COOP ? (int)(blockIdx.x % nb) : (int)blockIdx.x;
    const int g = COOP ? (int)(blockIdx.x / nb) : 0;      // my share of the scene
    const int nper = COOP ? (n_total + K - 1) / K : n_total;
    const int k_lo = g * nper;                             // first global index I own
    const int n = max(0, min(n_total, k_lo + nper) - k_lo);  // points I own (local indices 0..n-1)
    const float* __restrict__ xyz0 = xyz_all + (size_t)scene * n_total * 3;  // the scene
    const float* __restrict__ xyz = xyz0 + (size_t)k_lo * 3;                 // my range
    float* __restrict__ temp = temp_all + (size_t)scene * n_total + k_lo;
    int32_t* __restrict__ idx = idx_all + (size_t)scene * m;
    if (t == 0) failflag = 0;
    const int spin_limit = COOP ? g_fps_spin_limit : 0;

    uint32_t kT[P];   // tie-break value (0xffffffff = empty slot)
    int kk[P];
    float px[P], py[P], pz[P], tp[P];
    float blo[3], bhi[3];
    float lbest;
    int bi;
    fps_sorted_setup<P>(xyz, temp, n, k_lo, L, skey, red, px, py, pz, tp, kT, kk, blo, bhi, lbest, bi);

    // wave candidate cache (wave-uniform values)
    float wmax = -1.f, cx = 0.f, cy = 0.f, cz = 0.f;
    uint32_t wT = 0xffffffffu;
    bool dirty = true;  // no cached candidate yet
    bool publish = false;

    int old = 0;
    if (t == 0 && g == 0) idx[0] = 0;
    float x1 = xyz0[0], y1 = xyz0[1], z1 = xyz0[2];   // point 0 of the SCENE
    for (int j = 1; j < m; ++j) {
        // ---- conservative box test of my cluster against the new sample ----------------
        const float ex = fmaxf(fmaxf(blo[0] - x1, x1 - bhi[0]), 0.f);
        const float ey = fmaxf(fmaxf(blo[1] - y1, y1 - bhi[1]), 0.f);
        const float ez = fmaxf(fmaxf(blo[2] - z1, z1 - bhi[2]), 0.f);
        const float dbox = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
        const bool hit = dbox * 0.9999f < lbest;
#ifdef PDA_FPS_STATS
        if (lane == 0) atomicAdd(&g_fps_stats[1], 1ull);
#endif
        if (dirty || __ballot(hit) != 0ull) {
#ifdef PDA_FPS_STATS
            const unsigned long long hits_dbg = __ballot(hit);
            if (lane == 0) { atomicAdd(&g_fps_stats[0], 1ull); atomicAdd(&g_fps_stats[2], (unsigned long long)__builtin_popcountll(hits_dbg)); }
#endif
            // all P updates are independent; the arg-max is a tree (depth log2 P) that keeps the
            // LOWER slot on equal values (slots are in tie-break order) -- a serial chain of
            // compare/select would expose ~2 x P dependent VALU latencies on this lone wave
            float bv[P];
            int bx[P];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const float d = sqdist3(px[i], py[i], pz[i], x1, y1, z1);  // (x2 - x1), :133
                tp[i] = fminf(d, tp[i]);
                bv[i] = tp[i];
                bx[i] = i;
            }
#pragma unroll
            for (int st = 1; st < P; st <<= 1)
#pragma unroll
                for (int i = 0; i + st < P; i += 2 * st) {
                    const bool g = bv[i + st] > bv[i];  // strict: lower slot wins ties
                    bx[i] = g ? bx[i + st] : bx[i];
                    bv[i] = fmaxf(bv[i], bv[i + st]);
                }
            const float best = bv[0];
            const int b2 = bx[0];
            lbest = best; bi = b2;
            // wave candidate: max value, ties by tie-break value
            wmax = wave_max_f32(lbest);
            const unsigned long long eq = __ballot(lbest == wmax);
            int lstar;
            if (__builtin_popcountll(eq) == 1) {
                lstar = (int)__builtin_ctzll(eq);
            } else {
                uint32_t myT = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < P; ++i) myT = (bi == i) ? kT[i] : myT;
                const uint32_t tmin = wave_min_u32(lbest == wmax ? myT : 0xffffffffu);
                const unsigned long long eq2 = __ballot(lbest == wmax && myT == tmin);
                lstar = (int)__builtin_ctzll(eq2 | (1ull << 63));
            }
            lstar = __builtin_amdgcn_readfirstlane(lstar);
            const int js = __builtin_amdgcn_readlane(bi, lstar);
            fps_extract<0, P>(js, lstar, px, py, pz, kT, cx, cy, cz, wT);  // wave-uniform binary tree
            dirty = false;
            publish = true;
        }
        // ---- publish (only waves whose candidate changed), reduce in ONE wave, broadcast --------
        // Waves that skipped the scan have nothing new to say: their slot persists.  The 16-slot
        // reduction runs in wave 0 only and the winner goes through a 16-byte LDS record; the
        // other waves sleep at the barriers instead of issuing ~45 redundant instructions each
        // (4 waves share a SIMD: the redundant reductions were stealing issue slots from the one
        // wave that is actually scanning).
        if (publish && lane == 0) {
            slots[0][w] = make_uint2(__builtin_bit_cast(uint32_t, wmax), wT);
            cand[0][w] = make_float4(cx, cy, cz, 0.f);
        }
        publish = false;
        lds_barrier();
        if (w == 0) {
            uint2 sv = make_uint2(__builtin_bit_cast(uint32_t, -1.0f), 0xffffffffu);
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < FPS_WAVES) { sv = slots[0][lane]; c4 = cand[0][lane]; }
            const float v = __builtin_bit_cast(float, sv.x);
            const float bmax = row0_max_f32(v);
            const uint32_t bTw = row0_min_u32(v == bmax ? sv.y : 0xffffffffu);
            if (!COOP) {
                // exactly one slot holds (bmax, bT): that lane forwards its record
                if (lane < FPS_WAVES && v == bmax && sv.y == bTw)
                    cand[1][0] = make_float4(c4.x, c4.y, c4.z, __builtin_bit_cast(float, bTw));
            } else {
                const unsigned long long wm = __ballot(lane < FPS_WAVES && v == bmax && sv.y == bTw);
                const int lw = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(wm | (1ull << 63)) & (FPS_WAVES - 1));
                const uint32_t r0 = __builtin_bit_cast(uint32_t, bmax), r1 = bTw;
                const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane(__builtin_bit_cast(int, c4.x), lw);
                const uint32_t r3 = (uint32_t)__builtin_amdgcn_readlane(__builtin_bit_cast(int, c4.y), lw);
                const uint32_t r4 = (uint32_t)__builtin_amdgcn_readlane(__builtin_bit_cast(int, c4.z), lw);
                // publish my record: 5 granules {payload, tag}
                const uint32_t tag = (epoch << 17) | (uint32_t)j;  // 15-bit launch epoch | 17-bit round
                unsigned long long* base = g_fps_xbuf +
                    ((((size_t)(epoch % FPS_XBUF_REGIONS) * FPS_XBUF_SCENES + scene) * 2 + (j & 1)) * FPS_MAX_K) * 5;
                if (lane < 5) {
                    const uint32_t pay = lane == 0 ? r0 : (lane == 1 ? r1 : (lane == 2 ? r2 : (lane == 3 ? r3 : r4)));
                    __hip_atomic_store(base + g * 5 + lane, ((unsigned long long)tag << 32) | pay, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
                // collect the K records (lane 5q+f polls field f of workgroup q), bounded spin
                unsigned long long gv = 0;
                bool ok = lane >= 5 * K;
                bool failed = false;
                for (int spin = 0;; ++spin) {
                    if (!ok) {
                        gv = __hip_atomic_load(base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (uint32_t)(gv >> 32) == tag;
                    }
                    if (__ballot(!ok) == 0ull) break;
                    if (spin >= spin_limit) { failed = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                const int pay = (int)(uint32_t)gv;
                const int src = (5 * lane) & 63;
                const uint32_t f0 = (uint32_t)__shfl(pay, src), f1 = (uint32_t)__shfl(pay, (src + 1) & 63);
                const uint32_t f2 = (uint32_t)__shfl(pay, (src + 2) & 63), f3 = (uint32_t)__shfl(pay, (src + 3) & 63);
                const uint32_t f4 = (uint32_t)__shfl(pay, (src + 4) & 63);
                const float vq = lane < K ? __builtin_bit_cast(float, f0) : -1.0f;
                const uint32_t tq = lane < K ? f1 : 0xffffffffu;
                const float gmax = row0_max_f32(vq);
                const uint32_t gT = row0_min_u32(vq == gmax ? tq : 0xffffffffu);
                if (failed && lane == 0) {
                    failflag = 1;
                    __hip_atomic_store(&g_fps_fail[(epoch % FPS_XBUF_REGIONS) * FPS_XBUF_SCENES + scene], epoch,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicAdd(&g_fps_fail_total, 1ull);
                }
                if (lane < K && vq == gmax && tq == gT)
                    cand[1][0] = make_float4(__builtin_bit_cast(float, f2), __builtin_bit_cast(float, f3),
                                             __builtin_bit_cast(float, f4), __builtin_bit_cast(float, gT));
            }
        }
        lds_barrier();
        if (COOP && failflag) break;  // an exchange timed out: stop (bounded); fps_recover_kernel redoes this scene
        const float4 win = cand[1][0];
        x1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, win.x)));
        y1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, win.y)));
        z1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, win.z)));
        const uint32_t bT = (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, win.w));
        old = (int)fps_tiebreak_decode(bT, L);
        if (t == 0 && g == 0) idx[j] = old;
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
        if (kk[i] >= 0) temp[kk[i]] = tp[i];
}

// ---------------------------------------------------------------------------------------------
// fps_chain_kernel<P>: the pruned form with SEVERAL samples per synchronisation (2048 <= N <= 16384).
//
// FPS is m-1 dependent rounds, and a round of the pruned kernel is two barriers plus one wave scanning alone
// (0.91 us).  Most of that dependence is not real: the next sample is the best candidate that the new sample does
// not reach, and samples are far apart by construction.  Every row of 16 lanes (256 curve-consecutive points) keeps a
// record {best value, tie-break T, the best point's coordinates, second-best value}: 64 records per scene.  After
// ONE barrier wave 0 walks the records (one per lane):
//   * the best VALID record is the next sample if its value is above every bound of the invalidated rows;
//   * taking a sample invalidates its own row (whose new maximum is bounded by its second-best value) and every
//     row whose best point the sample reaches (d < value, the same f32 expression the scan evaluates, so the test
//     is exact; bound max(second, d));  all other records stay exact -- their points may change, their best cannot.
// The walk stops at the first candidate it cannot prove (or FPS_CHAIN_MAX); its samples go to LDS, and after the
// second barrier every wave applies the whole chain to the clusters each sample can reach (the conservative box
// test of the pruned kernel) and refreshes its four records once (row all-reduces by DPP row rotations; every lane
// selects its own best point, so there is no run-time register index and no scalar branch tree).  Indices and the
// final `temp` are bit-identical to the reference: skipped updates are no-ops, accepted samples are proven arg-maxima
// under the same total order.
// Measured (MI355X, 16384 -> 4096, B = 2, profiles/r02_fps_variants.txt): 6.65 samples per synchronisation (616
// instead of 4095; the host replay build/fps_chain_sim2.py predicts exactly that), 3.82 -> 2.46 ms.  Per
// synchronisation ~10k cycles: busiest wave's updates + refresh ~4.0k, the walk ~3.7k (a dependent chain of
// ~490 cycles per accepted sample on one wave), two barriers and their skew ~2.5k.  The walk's two reductions
// per sample (candidate, bounds of the reached rows) are one pass of two interleaved DPP chains since.
constexpr int FPS_CHAIN_MAX = 16;
constexpr int FPS_RECORDS = FPS_WAVES * 4;   // one record per row of 16 lanes (256 points)

template <int P>
__global__ __launch_bounds__(FPS_THREADS) void fps_chain_kernel(const float* __restrict__ xyz_all, float* __restrict__ temp_all,
                                                                 int32_t* __restrict__ idx_all, int n, int m, int L) {
    constexpr int NS = FPS_THREADS * P;
    __shared__ uint32_t skey[NS];
    __shared__ float red[FPS_WAVES * 6];
    __shared__ float4 rec_a[FPS_RECORDS];   // {value, T, x, y} of the best point of a row of 16 lanes (256 points)
    __shared__ float2 rec_b[FPS_RECORDS];   // {z, second-best value of the row}
    __shared__ float4 chain[FPS_CHAIN_MAX]; // {x, y, z, T} of the samples decided at the last synchronisation
    __shared__ int chain_n;
    const int t = threadIdx.x;
    const int lane = lane_id();
    const int w = wave_id();
    const float* __restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    float* __restrict__ temp = temp_all + (size_t)blockIdx.x * n;
    int32_t* __restrict__ idx = idx_all + (size_t)blockIdx.x * m;

    uint32_t kT[P];
    int kk[P];
    float px[P], py[P], pz[P], tp[P];
    float blo[3], bhi[3];
    float lbest;
    int bi;
    fps_sorted_setup<P>(xyz, temp, n, 0, L, skey, red, px, py, pz, tp, kT, kk, blo, bhi, lbest, bi);

    if (t == 0) {
        idx[0] = 0;
        chain[0] = make_float4(xyz[0], xyz[1], xyz[2], 0.f);   // sample 0 = point 0
        chain_n = 1;
    }
    __syncthreads();
#ifdef PDA_FPS_STATS
    unsigned long long t_scan = 0, t_dec = 0, t_bar = 0, n_round = 0, n_scanw = 0, t0s;
#define FPS_T0() t0s = __builtin_readcyclecounter()
#define FPS_T1(acc) acc += __builtin_readcyclecounter() - t0s
#else
#define FPS_T0()
#define FPS_T1(acc)
#endif
    int j = 1;            // indices decided so far
    int cn = 1;           // samples in the current chain: idx[j - cn .. j - 1]
    bool dirty = true;    // no record published yet
    while (true) {
        // ---- apply the chain: sample c is idx[j - cn + c]; the last index of all, idx[m - 1], is never applied ----
        const int napply = min(cn, (m - 1) - (j - cn));
        bool scanned = dirty;
        FPS_T0();
        for (int c = 0; c < napply; ++c) {
            const float4 s4 = chain[c];      // same address in every lane
            const float ex = fmaxf(fmaxf(blo[0] - s4.x, s4.x - bhi[0]), 0.f);
            const float ey = fmaxf(fmaxf(blo[1] - s4.y, s4.y - bhi[1]), 0.f);
            const float ez = fmaxf(fmaxf(blo[2] - s4.z, s4.z - bhi[2]), 0.f);
            const float dbox = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
            const bool hit = dbox * 0.9999f < lbest;     // lbest of the last refresh: an upper bound, still conservative
            if (__ballot(hit) != 0ull) {
#pragma unroll
                for (int i = 0; i < P; ++i) tp[i] = fminf(sqdist3(px[i], py[i], pz[i], s4.x, s4.y, s4.z), tp[i]);   // (x2 - x1), :133
                scanned = true;
            }
        }
        if (scanned) {
            // lane: best slot (lower slot wins ties: slots are in tie-break order) and second-best VALUE
            float bv[P], sv[P];
            int bx[P];
#pragma unroll
            for (int i = 0; i < P; ++i) { bv[i] = tp[i]; bx[i] = i; sv[i] = -1.f; }
#pragma unroll
            for (int st = 1; st < P; st <<= 1)
#pragma unroll
                for (int i = 0; i + st < P; i += 2 * st) {
                    const bool g = bv[i + st] > bv[i];
                    sv[i] = fmaxf(fmaxf(sv[i], sv[i + st]), fminf(bv[i], bv[i + st]));
                    bx[i] = g ? bx[i + st] : bx[i];
                    bv[i] = fmaxf(bv[i], bv[i + st]);
                }
            lbest = bv[0]; bi = bx[0];
            const float lsec = sv[0];
            // my best point (every lane selects its own: no run-time register index, no scalar branch tree)
            float mx = px[0], my = py[0], mz = pz[0];
            uint32_t mT = kT[0];
#pragma unroll
            for (int i = 1; i < P; ++i) {
                const bool sel = bi == i;
                mx = sel ? px[i] : mx; my = sel ? py[i] : my; mz = sel ? pz[i] : mz; mT = sel ? kT[i] : mT;
            }
            // one record per row of 16 lanes: the row's best (value, then tie-break), its second-best value (as a
            // multiset: a tie at the maximum makes it the maximum)
            const float rmax = row_allmax_f32(lbest);
            const uint32_t rT = row_allmin_u32(lbest == rmax ? mT : 0xffffffffu);
            const bool iswin = lbest == rmax && mT == rT;
            const float rsec = row_allmax_f32(iswin ? lsec : lbest);
            if (iswin) {       // an empty row (all lanes -1 / 0xffffffff) writes the same inert record from every lane
                rec_a[t >> 4] = make_float4(lbest, __builtin_bit_cast(float, mT), mx, my);
                rec_b[t >> 4] = make_float2(mz, rsec);
            }
            dirty = false;
#ifdef PDA_FPS_STATS
            n_scanw++;
#endif
        }
        FPS_T1(t_scan);
        if (j >= m) break;
        FPS_T0();
        lds_barrier();
        FPS_T1(t_bar);
        FPS_T0();
        if (w == 0) {
            // ---- the walk over the 64 records (one per lane); an invalid record carries value -1 ----
            const float4 ra = rec_a[lane];
            const float2 rb = rec_b[lane];
            float val = ra.x;
            const uint32_t rT = __builtin_bit_cast(uint32_t, ra.y);
            const float rx = ra.z, ry = ra.w, rz = rb.x, sec = rb.y;
            // bounds of the invalidated rows: `sbound` from the accepted samples' own rows (wave-uniform), `lbound` per lane
            // from the rows a sample reached -- reduced together with the candidate search of the NEXT step (one pass of
            // two interleaved DPP chains) instead of a reduction of its own behind every sample that reaches a row
            // Every value compared here is >= +0.0 or exactly -1.0f, so the bit patterns order like the numbers and the
            // wave-uniform part of the walk (bounds, stop test) runs on the scalar unit.
            int sbound = __builtin_bit_cast(int, -1.f);
            float lbound = -1.f;
            int ox = 0, oy = 0, oz = 0, oT = 0;     // lane c collects sample c: one LDS write and one index store at the end
            const int rem = min(FPS_CHAIN_MAX, m - j);
            int c = 0;
            while (c < rem) {
                float cv = val, mb = lbound;
                wave_max2_f32(cv, mb);
                const int cvi = __builtin_bit_cast(int, cv), mbi = __builtin_bit_cast(int, mb);
                if (cvi < 0 || max(sbound, mbi) >= cvi) break;       // nothing valid left / an invalidated row may hold more
                const unsigned long long eq = __ballot(val == cv);
                int cT, wl;
                if (__builtin_popcountll(eq) == 1) {
                    wl = (int)__builtin_ctzll(eq);
                    cT = __builtin_amdgcn_readlane((int)rT, wl);
                } else {
                    cT = (int)wave_min_u32(val == cv ? rT : 0xffffffffu);
                    wl = (int)__builtin_ctzll(__ballot(val == cv && rT == (uint32_t)cT) | (1ull << 63));
                }
                const int sxi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, rx), wl);
                const int syi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, ry), wl);
                const int szi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, rz), wl);
                sbound = max(sbound, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sec), wl));   // its own row: bounded by its second best
                const bool mine = lane == c;
                ox = mine ? sxi : ox; oy = mine ? syi : oy; oz = mine ? szi : oz; oT = mine ? cT : oT;
                // which other records does this sample invalidate?  Exactly those whose best point it reaches
                // (an invalid record has value -1: d >= 0 never reaches it again; the sample's own record goes first).
                val = writelane_minus_one(wl, val);
                const float d = sqdist3(rx, ry, rz, __builtin_bit_cast(float, sxi), __builtin_bit_cast(float, syi), __builtin_bit_cast(float, szi));
                const bool reach = d < val;
                lbound = reach ? fmaxf(lbound, fmaxf(sec, d)) : lbound;
                val = reach ? -1.f : val;
                ++c;
            }
            if (lane < c) {
                chain[lane] = make_float4(__builtin_bit_cast(float, ox), __builtin_bit_cast(float, oy), __builtin_bit_cast(float, oz),
                                          __builtin_bit_cast(float, oT));
                idx[j + lane] = (int)fps_tiebreak_decode((uint32_t)oT, L);
            }
            if (lane == 0) chain_n = c;
        }
        FPS_T1(t_dec);
#ifdef PDA_FPS_STATS
        n_round++;
#endif
        FPS_T0();
        lds_barrier();
        FPS_T1(t_bar);
        cn = __builtin_amdgcn_readfirstlane(chain_n);
        j += cn;
    }
#ifdef PDA_FPS_STATS
    if (lane == 0 && blockIdx.x == 0) {
        atomicAdd(&g_fps_stats[4], n_round); atomicAdd(&g_fps_stats[5], n_scanw);
        atomicMax(&g_fps_stats[6], t_scan); atomicAdd(&g_fps_stats[7], t_scan);
        if (w == 0) { atomicAdd(&g_fps_stats[8], t_dec); atomicAdd(&g_fps_stats[9], t_bar); }
        if (w == 5) atomicAdd(&g_fps_stats[10], t_bar);
    }
#endif
#pragma unroll
    for (int i = 0; i < P; ++i)
        if (kk[i] >= 0) temp[kk[i]] = tp[i];
}

// ---------------------------------------------------------------------------------------------
// fps_chain_coop_kernel<P>: the cooperative form (K workgroups per scene, 24576 < N <= 65536) with SEVERAL samples per
// exchange.  fps_pruned_kernel<P, true> exchanges one winner per round: 16383 rounds of ~2.1 us for 60000 -> 16384, the
// shipped ONCE input (34.5 ms: twice the rest of the training iteration).  Here every workgroup keeps the 64 row records
// of fps_chain_kernel for its own quarter of the cloud, publishes all of them once per synchronisation (6 x 64 tagged
// 8-byte granules), collects the 64 K records of the scene, and wave 0 of EVERY workgroup walks them -- the same data and
// the same deterministic instructions in each, so all K workgroups decide the same chain without a second exchange.  The
// acceptance rule and its proof are fps_chain_kernel's (a record = any set of points with its best point, best value and
// second-best value; which workgroup owns the row does not matter); the exchange, its bounded polls and the failure model
// are fps_pruned_kernel<P, true>'s (a timeout marks the scene, every workgroup stops, fps_recover_kernel recomputes it).
// Indices and the final temp are bit-identical to the reference.  Measured: 60000 -> 16384, 2 scenes 34.5 -> 12.5 ms.  Dead ends:
// the K workgroups of a scene placed on ONE XCD (blockIdx = 8 (K j + g) + s): 12.43 against 12.58 ms -- the agent-scope granules
// travel through memory either way; K = 2 / 4 workgroups on 16384 points (PDA_FPS_COOP_KMIN): 3.8-3.9 ms against
// fps_chain_kernel's 2.40 -- an exchange costs more than the smaller shares save.
#ifndef PDA_FPS_CC_CHAIN_MAX
#define PDA_FPS_CC_CHAIN_MAX 32
#endif
constexpr int FPS_CC_CHAIN_MAX = PDA_FPS_CC_CHAIN_MAX;     // samples per exchange at most (a lane of the walk collects one: <= 64)
constexpr int FPS_CC_WORDS = 6;                   // value, T, x, y, z, second-best value
constexpr int FPS_CC_GRANULES = FPS_RECORDS * FPS_CC_WORDS;      // per workgroup and synchronisation
__device__ unsigned long long g_fps_cbuf[FPS_XBUF_REGIONS * FPS_XBUF_SCENES * 2 * FPS_MAX_K * FPS_CC_GRANULES];

template <int P>
__global__ __launch_bounds__(FPS_THREADS) void fps_chain_coop_kernel(const float* __restrict__ xyz_all, float* __restrict__ temp_all,
                                                                      int32_t* __restrict__ idx_all, int n_total, int m, int L,
                                                                      int K, int nb, uint32_t epoch) {
    constexpr int NS = FPS_THREADS * P;
    __shared__ uint32_t skey[NS];            // the sort's keys; behind the set-up: the records of all K workgroups
    __shared__ float red[FPS_WAVES * 6];
    __shared__ uint32_t rec[FPS_CC_GRANULES];   // my 64 records, [row][word]
    __shared__ float4 chain[FPS_CC_CHAIN_MAX];
    __shared__ int chain_n, failflag;
    const int t = threadIdx.x;
    const int lane = lane_id();
    const int w = wave_id();
    const int scene = (int)(blockIdx.x % nb), g = (int)(blockIdx.x / nb);
    const int nper = (n_total + K - 1) / K;
    const int k_lo = g * nper;
    const int n = max(0, min(n_total, k_lo + nper) - k_lo);
    const float* __restrict__ xyz0 = xyz_all + (size_t)scene * n_total * 3;
    const float* __restrict__ xyz = xyz0 + (size_t)k_lo * 3;
    float* __restrict__ temp = temp_all + (size_t)scene * n_total + k_lo;
    int32_t* __restrict__ idx = idx_all + (size_t)scene * m;
    const int spin_limit = g_fps_spin_limit;

    uint32_t kT[P];
    float px[P], py[P], pz[P], tp[P];
    float blo[3], bhi[3];
    float lbest;
    int bi;
    uint16_t* const kk_lds = reinterpret_cast<uint16_t*>(skey + NS / 2);     // upper half of skey, behind the set-up
    {
        int kk[P];
        fps_sorted_setup<P>(xyz, temp, n, k_lo, L, skey, red, px, py, pz, tp, kT, kk, blo, bhi, lbest, bi);
        __syncthreads();                      // everyone is done with skey
        // the points' local indices are needed again only for the final store of temp: 16 registers a lane does not have to
        // carry through 1600 synchronisations (the walk's 24 record registers come on top of the 96 of the points)
#pragma unroll
        for (int i = 0; i < P; ++i) kk_lds[i * FPS_THREADS + t] = (uint16_t)(kk[i] >= 0 ? kk[i] : 0xffff);
    }
    uint32_t* const allrec = skey;            // [workgroup q][row][word]: 6 K x 256 bytes at the bottom of skey

    if (t == 0) {
        if (g == 0) idx[0] = 0;
        chain[0] = make_float4(xyz0[0], xyz0[1], xyz0[2], 0.f);   // sample 0 = point 0 of the SCENE
        chain_n = 1;
        failflag = 0;
    }
    if (t < FPS_CC_GRANULES) rec[t] = (t % FPS_CC_WORDS == 1) ? 0xffffffffu : __builtin_bit_cast(uint32_t, -1.f);   // inert until refreshed
    __syncthreads();
    int j = 1, cn = 1;
    bool dirty = true;
    uint32_t sync = 0;
    while (true) {
        // ---- apply the chain to my points (fps_chain_kernel's phase, unchanged) ----
        const int napply = min(cn, (m - 1) - (j - cn));
        bool scanned = dirty;
        for (int c = 0; c < napply; ++c) {
            const float4 s4 = chain[c];
            const float ex = fmaxf(fmaxf(blo[0] - s4.x, s4.x - bhi[0]), 0.f);
            const float ey = fmaxf(fmaxf(blo[1] - s4.y, s4.y - bhi[1]), 0.f);
            const float ez = fmaxf(fmaxf(blo[2] - s4.z, s4.z - bhi[2]), 0.f);
            const float dbox = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
            const bool hit = dbox * 0.9999f < lbest;
            if (__ballot(hit) != 0ull) {
#pragma unroll
                for (int i = 0; i < P; ++i) tp[i] = fminf(sqdist3(px[i], py[i], pz[i], s4.x, s4.y, s4.z), tp[i]);
                scanned = true;
            }
        }
        if (scanned) {
            float bv[P], sv[P];
            int bx[P];
#pragma unroll
            for (int i = 0; i < P; ++i) { bv[i] = tp[i]; bx[i] = i; sv[i] = -1.f; }
#pragma unroll
            for (int st = 1; st < P; st <<= 1)
#pragma unroll
                for (int i = 0; i + st < P; i += 2 * st) {
                    const bool gt = bv[i + st] > bv[i];
                    sv[i] = fmaxf(fmaxf(sv[i], sv[i + st]), fminf(bv[i], bv[i + st]));
                    bx[i] = gt ? bx[i + st] : bx[i];
                    bv[i] = fmaxf(bv[i], bv[i + st]);
                }
            lbest = bv[0]; bi = bx[0];
            const float lsec = sv[0];
            float mx = px[0], my = py[0], mz = pz[0];
            uint32_t mT = kT[0];
#pragma unroll
            for (int i = 1; i < P; ++i) {
                const bool sel = bi == i;
                mx = sel ? px[i] : mx; my = sel ? py[i] : my; mz = sel ? pz[i] : mz; mT = sel ? kT[i] : mT;
            }
            const float rmax = row_allmax_f32(lbest);
            const uint32_t rT = row_allmin_u32(lbest == rmax ? mT : 0xffffffffu);
            const bool iswin = lbest == rmax && mT == rT;
            const float rsec = row_allmax_f32(iswin ? lsec : lbest);
            if (iswin) {
                uint32_t* r = rec + (t >> 4) * FPS_CC_WORDS;
                r[0] = __builtin_bit_cast(uint32_t, lbest); r[1] = mT;
                r[2] = __builtin_bit_cast(uint32_t, mx); r[3] = __builtin_bit_cast(uint32_t, my); r[4] = __builtin_bit_cast(uint32_t, mz);
                r[5] = __builtin_bit_cast(uint32_t, rsec);
            }
            dirty = false;
        }
        if (j >= m) break;
        lds_barrier();
        // ---- exchange: publish my 64 records, collect the scene's 64 K ----
        {
            const uint32_t tag = (epoch << 17) | (sync & 0x1ffffu);
            unsigned long long* base = g_fps_cbuf +
                (((size_t)(epoch % FPS_XBUF_REGIONS) * FPS_XBUF_SCENES + scene) * 2 + (sync & 1)) * FPS_MAX_K * FPS_CC_GRANULES;
            if (t < FPS_CC_GRANULES)
                __hip_atomic_store(base + g * FPS_CC_GRANULES + t, ((unsigned long long)tag << 32) | rec[t], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            const int total = K * FPS_CC_GRANULES;
            unsigned long long g0 = 0, g1 = 0;
            bool ok0 = t >= total, ok1 = t + FPS_THREADS >= total;
            bool failed = false;
            for (int spin = 0;; ++spin) {
                if (!ok0) { g0 = __hip_atomic_load(base + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok0 = (uint32_t)(g0 >> 32) == tag; }
                if (!ok1) { g1 = __hip_atomic_load(base + t + FPS_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok1 = (uint32_t)(g1 >> 32) == tag; }
                if (__ballot(!ok0 || !ok1) == 0ull) break;
                if (spin >= spin_limit) { failed = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (t < total) allrec[t] = (uint32_t)g0;
            if (t + FPS_THREADS < total) allrec[t + FPS_THREADS] = (uint32_t)g1;
            if (failed && lane == 0) {
                failflag = 1;
                __hip_atomic_store(&g_fps_fail[(epoch % FPS_XBUF_REGIONS) * FPS_XBUF_SCENES + scene], epoch, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&g_fps_fail_total, 1ull);
            }
        }
        lds_barrier();
        if (failflag) break;       // an exchange timed out: stop (bounded); fps_recover_kernel redoes this scene
        // ---- the walk: wave 0 of every workgroup, over the same 64 K records (slot q of a lane = row `lane` of workgroup q) ----
        if (w == 0) {
            float val[FPS_MAX_K], rx[FPS_MAX_K], ry[FPS_MAX_K], rz[FPS_MAX_K], sec[FPS_MAX_K];
            uint32_t rT[FPS_MAX_K];
#pragma unroll
            for (int q = 0; q < FPS_MAX_K; ++q) {
                const uint32_t* r = allrec + (q * FPS_RECORDS + lane) * FPS_CC_WORDS;
                const bool live = q < K;
                val[q] = live ? __builtin_bit_cast(float, r[0]) : -1.f;
                rT[q] = live ? r[1] : 0xffffffffu;
                rx[q] = live ? __builtin_bit_cast(float, r[2]) : 0.f;
                ry[q] = live ? __builtin_bit_cast(float, r[3]) : 0.f;
                rz[q] = live ? __builtin_bit_cast(float, r[4]) : 0.f;
                sec[q] = live ? __builtin_bit_cast(float, r[5]) : -1.f;
            }
            int sbound = __builtin_bit_cast(int, -1.f);
            float lbound = -1.f;
            int ox = 0, oy = 0, oz = 0, oT = 0;
            const int rem = min(FPS_CC_CHAIN_MAX, m - j);
            int c = 0;
            while (c < rem) {
                float cv = fmaxf(fmaxf(val[0], val[1]), fmaxf(val[2], val[3])), mb = lbound;
                wave_max2_f32(cv, mb);
                const int cvi = __builtin_bit_cast(int, cv), mbi = __builtin_bit_cast(int, mb);
                if (cvi < 0 || max(sbound, mbi) >= cvi) break;
                // my best tie-break value among the slots that hold the maximum, and that slot's record
                uint32_t lT = 0xffffffffu;
                float lx = 0.f, ly = 0.f, lz = 0.f, ls = -1.f;
#pragma unroll
                for (int q = 0; q < FPS_MAX_K; ++q) {
                    const bool better = val[q] == cv && rT[q] < lT;
                    lT = better ? rT[q] : lT; lx = better ? rx[q] : lx; ly = better ? ry[q] : ly; lz = better ? rz[q] : lz; ls = better ? sec[q] : ls;
                }
                const uint32_t cT = wave_min_u32(lT);
                const int wl = (int)__builtin_ctzll(__ballot(lT == cT) | (1ull << 63));
                const int sxi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, lx), wl);
                const int syi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, ly), wl);
                const int szi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, lz), wl);
                sbound = max(sbound, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls), wl));
                const bool mine = lane == c;
                ox = mine ? sxi : ox; oy = mine ? syi : oy; oz = mine ? szi : oz; oT = mine ? (int)cT : oT;
                const float sx = __builtin_bit_cast(float, sxi), sy = __builtin_bit_cast(float, syi), sz = __builtin_bit_cast(float, szi);
#pragma unroll
                for (int q = 0; q < FPS_MAX_K; ++q) {
                    if (lane == wl && val[q] == cv && rT[q] == cT) val[q] = -1.f;        // the sample's own record goes first
                    const float d = sqdist3(rx[q], ry[q], rz[q], sx, sy, sz);
                    const bool reach = d < val[q];
                    lbound = reach ? fmaxf(lbound, fmaxf(sec[q], d)) : lbound;
                    val[q] = reach ? -1.f : val[q];
                }
                ++c;
            }
            if (lane < c) {
                chain[lane] = make_float4(__builtin_bit_cast(float, ox), __builtin_bit_cast(float, oy), __builtin_bit_cast(float, oz),
                                          __builtin_bit_cast(float, oT));
                if (g == 0) idx[j + lane] = (int)fps_tiebreak_decode((uint32_t)oT, L);
            }
            if (lane == 0) chain_n = c;
        }
        lds_barrier();
        cn = __builtin_amdgcn_readfirstlane(chain_n);
        j += cn;
        ++sync;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = kk_lds[i * FPS_THREADS + t];
        if (k != 0xffff) temp[k] = tp[i];
    }
}

template <bool WITH_DIST>
__device__ __forceinline__ void fps_stream_scene(const float* __restrict__ data, float* __restrict__ temp,
                                                 int32_t* __restrict__ idx, int n, int m, int L, uint2 (*slots)[FPS_WAVES]) {
    const int t = threadIdx.x;
    const int lane = lane_id();
    const int w = wave_id();
    const int nwaves = (int)(blockDim.x >> 6);
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

template <bool WITH_DIST>
__global__ __launch_bounds__(FPS_THREADS) void fps_stream_kernel(const float* __restrict__ data_all,
                                                                  float* __restrict__ temp_all,
                                                                  int32_t* __restrict__ idx_all, int n,
                                                                  int m, int L) {
    __shared__ uint2 slots[2][FPS_WAVES];
    fps_stream_scene<WITH_DIST>(data_all + (size_t)blockIdx.x * n * (WITH_DIST ? (size_t)n : 3),
                                temp_all + (size_t)blockIdx.x * n, idx_all + (size_t)blockIdx.x * m, n, m, L, slots);
}

// Behind every cooperative launch, same stream, one workgroup per scene: returns at once unless an exchange of
// that launch timed out for its scene; then it recomputes the scene from scratch with the streaming algorithm.
// It also wipes the exchange granules the launch used for its scene (`xbuf`: `xcount` granules per scene of this region):
// every workgroup of the cooperative launch has finished by now, so no reader can miss a record, and no granule keeps a
// tag that a launch 32767 epochs later -- same 15-bit epoch, same region, same round numbers -- would take for its own.
__global__ __launch_bounds__(FPS_THREADS) void fps_recover_kernel(const float* __restrict__ xyz_all,
                                                                   float* __restrict__ temp_all,
                                                                   int32_t* __restrict__ idx_all, int n, int m,
                                                                   int L, uint32_t epoch, unsigned long long* __restrict__ xbuf,
                                                                   int xcount) {
    __shared__ uint2 slots[2][FPS_WAVES];
    {
        unsigned long long* mine = xbuf + ((size_t)(epoch % FPS_XBUF_REGIONS) * FPS_XBUF_SCENES + blockIdx.x) * xcount;
        for (int i = threadIdx.x; i < xcount; i += FPS_THREADS)
            __hip_atomic_store(mine + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned int* mark = &g_fps_fail[(epoch % FPS_XBUF_REGIONS) * FPS_XBUF_SCENES + blockIdx.x];
    if (__hip_atomic_load(mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) return;   // wave-uniform
    float* __restrict__ temp = temp_all + (size_t)blockIdx.x * n;
    for (int k = threadIdx.x; k < n; k += FPS_THREADS) temp[k] = 1e10f;   // the entry contract (pointnet2_utils.py:26);
    // thread t re-reads only the elements it wrote itself (same stride), so no barrier is needed before the scan
    fps_stream_scene<false>(xyz_all + (size_t)blockIdx.x * n * 3, temp, idx_all + (size_t)blockIdx.x * m, n, m, L, slots);
    if (threadIdx.x == 0) __hip_atomic_store(mark, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// pointnet2_stack: stack_farthest_point_sampling_kernel (pointnet2_stack/src/sampling_gpu.cu:188-318).  Per-scene
// n / m come from device arrays, the block size is 1024 whatever n (:340, so the tie-break uses L = 10) and
// the indices written are global (local + scene start).  Streams xyz / temp like fps_stream_kernel.
__global__ __launch_bounds__(FPS_THREADS) void fps_stack_kernel(const float* __restrict__ xyz_all, float* __restrict__ temp_all,
                                                                const int32_t* __restrict__ xyz_batch_cnt,
                                                                int32_t* __restrict__ idx_all,
                                                                const int32_t* __restrict__ num_sampled) {
    __shared__ uint2 slots[2][FPS_WAVES];
    const int t = threadIdx.x, lane = lane_id(), w = wave_id();
    int start = 0, istart = 0;
    for (int k = 0; k < (int)blockIdx.x; ++k) { start += xyz_batch_cnt[k]; istart += num_sampled[k]; }
    const int n = xyz_batch_cnt[blockIdx.x], m = num_sampled[blockIdx.x];
    if (m <= 0) return;   // the reference stores idxs[0] even then, i.e. into the next scene's slot or past the end
    const float* __restrict__ data = xyz_all + (size_t)start * 3;
    float* __restrict__ temp = temp_all + start;
    int32_t* __restrict__ idx = idx_all + istart;
    constexpr int L = 10;
    int old = 0;
    if (t == 0) idx[0] = start;
    for (int j = 1; j < m; ++j) {
        const float x1 = data[old * 3 + 0], y1 = data[old * 3 + 1], z1 = data[old * 3 + 2];
        float best = -1.f;
        int bk = 0;
        for (int k = t; k < n; k += FPS_THREADS) {
            const float d = sqdist3(data[k * 3 + 0], data[k * 3 + 1], data[k * 3 + 2], x1, y1, z1);
            const float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            const bool g = d2 > best;
            bk = g ? k : bk;
            best = g ? d2 : best;
        }
        const uint32_t T = best >= 0.f ? fps_tiebreak((uint32_t)bk, L) : 0xffffffffu;
        old = fps_block_argmax(best, T, slots[j & 1], FPS_WAVES, w, lane, L);
        if (t == 0) idx[j] = old + start;
    }
}

static int ilog2(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}

// Workgroups of fps_pruned_kernel<16, true> the current device keeps resident at once.
static int coop_resident_workgroups() {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_pruned_kernel<16, true>, FPS_THREADS, 0) != hipSuccess) return 0;
    return cus * std::min(per_cu, 1);
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
    static const int no_prune = getenv("PDA_FPS_NO_PRUNE") ? atoi(getenv("PDA_FPS_NO_PRUNE")) : 0;
    static const int no_chain = getenv("PDA_FPS_NO_CHAIN") ? atoi(getenv("PDA_FPS_NO_CHAIN")) : 0;
    static const int coop_kmin = getenv("PDA_FPS_COOP_KMIN") ? atoi(getenv("PDA_FPS_COOP_KMIN")) : 1;     // experiments: more workgroups per scene
    if (!no_prune && !no_chain && n >= 2048 && n <= 16384 && m > 2 && coop_kmin <= 1) {
        if (P <= 2) hipLaunchKernelGGL((fps_chain_kernel<2>), grid, block, 0, stream, data, temp, idx, n, m, L);
        else if (P <= 4) hipLaunchKernelGGL((fps_chain_kernel<4>), grid, block, 0, stream, data, temp, idx, n, m, L);
        else if (P <= 8) hipLaunchKernelGGL((fps_chain_kernel<8>), grid, block, 0, stream, data, temp, idx, n, m, L);
        else hipLaunchKernelGGL((fps_chain_kernel<16>), grid, block, 0, stream, data, temp, idx, n, m, L);
        return check_launch(what);
    }
    if (!no_prune && n >= 2048 && n <= 16384 && m > 2 && coop_kmin <= 1) {
        if (P <= 2) hipLaunchKernelGGL((fps_pruned_kernel<2, false>), grid, block, 0, stream, data, temp, idx, n, m, L, 1, b, 0u);
        else if (P <= 4) hipLaunchKernelGGL((fps_pruned_kernel<4, false>), grid, block, 0, stream, data, temp, idx, n, m, L, 1, b, 0u);
        else if (P <= 8) hipLaunchKernelGGL((fps_pruned_kernel<8, false>), grid, block, 0, stream, data, temp, idx, n, m, L, 1, b, 0u);
        else hipLaunchKernelGGL((fps_pruned_kernel<16, false>), grid, block, 0, stream, data, temp, idx, n, m, L, 1, b, 0u);
        return check_launch(what);
    }
    static const int no_coop = getenv("PDA_FPS_NO_COOP") ? atoi(getenv("PDA_FPS_NO_COOP")) : 0;
    // every n above one workgroup's 16384 points (the register kernel held 16385...24576: 9.3 ms against 5.3 at 24576 -> 6144)
    static const int coop_from = getenv("PDA_FPS_COOP_FROM") ? atoi(getenv("PDA_FPS_COOP_FROM")) : 16384;
    if (!no_prune && !no_coop && n > coop_from && (n > 16384 || coop_kmin > 1) && n <= 16384 * FPS_MAX_K && m > 2 && m < (1 << 17)) {
        // K workgroups per scene, all resident together: a launch holds at most what the device admits at once
        // (one 1024-lane workgroup per CU for this kernel; the occupancy query only confirms >= 1) and 64 scenes
        static std::atomic<uint32_t> epoch_counter{1};
        static PerDevice<int> resident_of;
        const int resident = resident_of.get(coop_resident_workgroups);
        const int K = std::min(FPS_MAX_K, std::max(divup(n, 16384), coop_kmin));
        const int chunk = std::min(FPS_XBUF_SCENES, resident / K);
        if (chunk >= 1) {
            for (int s0 = 0; s0 < b; s0 += chunk) {
                const int nb = std::min(chunk, b - s0);
                uint32_t epoch = epoch_counter.fetch_add(1) & 0x7fffu;
                if (epoch == 0) epoch = epoch_counter.fetch_add(1) & 0x7fffu;   // 0 = "no failure" in g_fps_fail
                // several samples per exchange (fps_chain_coop_kernel); PDA_FPS_COOP_CHAIN=0: one winner per round
                static const bool coop_chain = !(getenv("PDA_FPS_COOP_CHAIN") && atoi(getenv("PDA_FPS_COOP_CHAIN")) == 0);
                if (coop_chain)
                    hipLaunchKernelGGL((fps_chain_coop_kernel<16>), dim3(nb * K), block, 0, stream,
                                       data + (size_t)s0 * n * 3, temp + (size_t)s0 * n, idx + (size_t)s0 * m, n, m, L, K, nb, epoch);
                else
                    hipLaunchKernelGGL((fps_pruned_kernel<16, true>), dim3(nb * K), block, 0, stream,
                                       data + (size_t)s0 * n * 3, temp + (size_t)s0 * n, idx + (size_t)s0 * m, n, m, L, K, nb, epoch);
                static PerDevice<unsigned long long*> cbuf_of, xbuf_of;
                unsigned long long* xbuf = coop_chain
                    ? cbuf_of.get([] { void* q = nullptr; return hipGetSymbolAddress(&q, HIP_SYMBOL(g_fps_cbuf)) == hipSuccess ? (unsigned long long*)q : (unsigned long long*)nullptr; })
                    : xbuf_of.get([] { void* q = nullptr; return hipGetSymbolAddress(&q, HIP_SYMBOL(g_fps_xbuf)) == hipSuccess ? (unsigned long long*)q : (unsigned long long*)nullptr; });
                const int xcount = 2 * FPS_MAX_K * (coop_chain ? FPS_CC_GRANULES : 5);      // both parities, all K slots
                PDA_REQUIRE(xbuf != nullptr, "%s: exchange buffer symbol", what);
                hipLaunchKernelGGL(fps_recover_kernel, dim3(nb), block, 0, stream, data + (size_t)s0 * n * 3,
                                   temp + (size_t)s0 * n, idx + (size_t)s0 * m, n, m, L, epoch, xbuf, xcount);
            }
            return check_launch(what);
        }   // a device too small to hold the K workgroups of one scene: streaming kernel below
    }
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

#ifdef PDA_FPS_STATS
PDA_API int pda_debug_fps_stats(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pda::g_fps_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(pda::g_fps_stats), z, sizeof(z)); }
    return 0;
}
#endif

PDA_API int pda_fps_coop_timeouts(unsigned long long* total, int reset) {
    PDA_REQUIRE(total != nullptr, "pda_fps_coop_timeouts: null pointer");
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpyFromSymbol(total, HIP_SYMBOL(pda::g_fps_fail_total), sizeof(unsigned long long)) != hipSuccess) {
        pda::set_error("pda_fps_coop_timeouts: %s", hipGetErrorString(hipGetLastError()));
        return PDA_ERR_LAUNCH;
    }
    if (reset) {
        const unsigned long long z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(pda::g_fps_fail_total), &z, sizeof(z)) != hipSuccess) {
            pda::set_error("pda_fps_coop_timeouts: reset failed: %s", hipGetErrorString(hipGetLastError()));
            return PDA_ERR_LAUNCH;
        }
    }
    return PDA_OK;
}

// number of non-zero exchange granules (both buffers) after everything queued has run: 0 when every launch was wiped
PDA_API long long pda_debug_fps_exchange_nonzero(void) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    static unsigned long long host[sizeof(pda::g_fps_cbuf) / sizeof(unsigned long long)];
    long long count = 0;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(pda::g_fps_cbuf), sizeof(pda::g_fps_cbuf)) != hipSuccess) return -1;
    for (size_t i = 0; i < sizeof(pda::g_fps_cbuf) / sizeof(unsigned long long); ++i) count += host[i] != 0;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(pda::g_fps_xbuf), sizeof(pda::g_fps_xbuf)) != hipSuccess) return -1;
    for (size_t i = 0; i < sizeof(pda::g_fps_xbuf) / sizeof(unsigned long long); ++i) count += host[i] != 0;
    return count;
}

PDA_API int pda_debug_fps_spin_limit(int polls) {
    const int v = polls < 0 ? pda::FPS_SPIN_LIMIT : polls;
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(pda::g_fps_spin_limit), &v, sizeof(v)) != hipSuccess) {
        pda::set_error("pda_debug_fps_spin_limit: %s", hipGetErrorString(hipGetLastError()));
        return PDA_ERR_LAUNCH;
    }
    return PDA_OK;
}

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

PDA_API int pda_stack_furthest_point_sampling(const float* xyz, float* temp, const int32_t* xyz_batch_cnt, int32_t* idx,
                                              const int32_t* num_sampled_points, int b, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0, "pda_stack_furthest_point_sampling: b=%d", b);
    if (b == 0) return PDA_OK;
    PDA_REQUIRE(xyz && temp && xyz_batch_cnt && idx && num_sampled_points, "pda_stack_furthest_point_sampling: null pointer");
    hipLaunchKernelGGL(pda::fps_stack_kernel, dim3(b), dim3(pda::FPS_THREADS), 0, (hipStream_t)stream, xyz, temp,
                       xyz_batch_cnt, idx, num_sampled_points);
    return pda::check_launch("pda_stack_furthest_point_sampling");
}
