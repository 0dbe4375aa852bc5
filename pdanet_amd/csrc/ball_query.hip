// ball_query.hip -- ball query for gfx950, index-exact w.r.t. the reference kernels
// ball_query_kernel_fast / ball_query_dilated_kernel_fast
// (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:9-45, :70-117).
//
// Design (not the reference's one-thread-per-centre scan of global memory):
//   * one LANE per centre, 64 centres per wave; the point being tested is the same for all
//     lanes, so the point stream is wave-uniform: it is fetched with SCALAR loads into SGPRs
//     (8 points = 96 B per batch) and the 7 VALU ops of a distance test take the point as an
//     SGPR operand -- no LDS staging, no VGPRs and no vector-memory traffic for xyz at all;
//   * the N points are split into S contiguous segments, one per wave of the workgroup, so a
//     64-centre tile is scanned by S waves in parallel (fills the chip at batch 2, where
//     there are only M/64*B tiles); each wave appends hits, in ascending index, to its own
//     per-centre partial row in LDS with a branch-free predicated ds_write;
//   * after a workgroup barrier the S partial rows of a centre are concatenated in segment
//     order (= ascending point index), truncated to nsample, padded with the first hit
//     (the reference's pre-fill, ball_query_gpu.cu:35-39) and written with coalesced stores;
//     rows with no hit are left untouched, as in the reference (:34 never true);
//   * up to 3 radii share one pass over the points (pda_ball_query_multi).
// Algorithmic HBM bytes per (scene, radius): ceil(M/256)*N*12 + M*12 + M*ns*4;
// the kernel is VALU-bound (M*N distance tests), see DESIGN.md.
#include "pda_common.h"

#include <algorithm>

namespace pda {

constexpr int BQ_BATCH = 8;
constexpr int BQ_MAX_NR = 3;

struct BqParams {
    const float* new_xyz;
    const float* xyz;
    int32_t* idx[BQ_MAX_NR];
    float r2[BQ_MAX_NR];   // radius*radius computed in float (ball_query_gpu.cu:23)
    float r2min;           // dilated only: min_radius^2 (:85)
    int ns[BQ_MAX_NR];
    int lds_row_off[BQ_MAX_NR];  // int offset of radius i's partial rows
    int lds_cnt_off[BQ_MAX_NR];  // int offset of radius i's counts
    int n, m, seglen;
};

// RowT: element type of the partial rows in LDS.  uint16_t (index relative to the segment start)
// whenever a segment has < 65536 points: halves the LDS per wave, i.e. doubles the waves a CU holds.
template <int NR, bool DILATED, typename RowT>
__global__ __launch_bounds__(1024) void ball_query_kernel(const BqParams p) {
    extern __shared__ int32_t lds[];
    RowT* const rows = reinterpret_cast<RowT*>(lds);   // rows first, int32 counts after (lds_cnt_off)
    const int w = wave_id();
    const int lane = lane_id();
    const int S = (int)(blockDim.x >> 6);
    const int bs = blockIdx.y;
    const int tile0 = blockIdx.x * PDA_WAVE;
    const int c = tile0 + lane;
    const bool valid = c < p.m;

    // lanes past M load the last centre (no divergent branch here: it would make the
    // point-stream base look divergent to the compiler) and are masked through `valid`
    const float* q = p.new_xyz + ((size_t)bs * p.m + min(c, p.m - 1)) * 3;
    const float cx = q[0], cy = q[1], cz = q[2];
    const cfloat_ptr pts = as_constant(uniform_ptr(p.xyz + (size_t)bs * p.n * 3));

    int cnt[NR], row[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        cnt[i] = valid ? 0 : p.ns[i];  // lanes past M behave as full rows
        row[i] = p.lds_row_off[i] + (w * PDA_WAVE + lane) * (p.ns[i] + 1);
    }

    // radii need not be sorted: the batch-level "did anybody hit" test uses the largest
    float rmax = p.r2[0];
#pragma unroll
    for (int i = 1; i < NR; ++i) rmax = fmaxf(rmax, p.r2[i]);

    const int k_begin = w * p.seglen;
    const int k_end = min(p.n, k_begin + p.seglen);

    for (int k0 = k_begin; k0 < k_end; k0 += BQ_BATCH) {
        // wave-uniform fetch of BQ_BATCH points (scalar loads); the tail batch clamps the
        // point index so nothing past xyz[n-1] is read, its hits are masked by k < k_end.
        float px[BQ_BATCH], py[BQ_BATCH], pz[BQ_BATCH];
        if (k0 + BQ_BATCH <= p.n) {
#pragma unroll
            for (int u = 0; u < BQ_BATCH; ++u) {
                px[u] = pts[(k0 + u) * 3 + 0];
                py[u] = pts[(k0 + u) * 3 + 1];
                pz[u] = pts[(k0 + u) * 3 + 2];
            }
        } else {
#pragma unroll
            for (int u = 0; u < BQ_BATCH; ++u) {
                const int k = min(k0 + u, p.n - 1);
                px[u] = pts[k * 3 + 0];
                py[u] = pts[k * 3 + 1];
                pz[u] = pts[k * 3 + 2];
            }
        }
        float d2[BQ_BATCH];
        unsigned long long anyhit = 0ull;  // OR of the per-point hit masks, kept in SGPRs
#pragma unroll
        for (int u = 0; u < BQ_BATCH; ++u) {
            d2[u] = sqdist3(cx, cy, cz, px[u], py[u], pz[u]);  // (new - x), ball_query_gpu.cu:33
            anyhit |= __ballot(DILATED ? (d2[u] < rmax || d2[u] == 0.f) : (d2[u] < rmax));
        }
        if (anyhit == 0ull) continue;  // wave-uniform: nobody hit anything in this batch

        bool room = false;
#pragma unroll
        for (int u = 0; u < BQ_BATCH; ++u) {
            const int k = k0 + u;
            const bool in_seg = k < k_end;
            if (DILATED) {
                // ball_query_gpu.cu:96-115: d2 == 0 appends, then the shell test appends
                const bool t0 = in_seg && (d2[u] == 0.f) && cnt[0] < p.ns[0];
                rows[t0 ? row[0] + cnt[0] : row[0] + p.ns[0]] = (RowT)(k - k_begin);
                cnt[0] += t0 ? 1 : 0;
                const bool t1 = in_seg && (d2[u] >= p.r2min) && (d2[u] < p.r2[0]) && cnt[0] < p.ns[0];
                rows[t1 ? row[0] + cnt[0] : row[0] + p.ns[0]] = (RowT)(k - k_begin);
                cnt[0] += t1 ? 1 : 0;
            } else {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const bool t = in_seg && (d2[u] < p.r2[i]) && cnt[i] < p.ns[i];
                    rows[t ? row[i] + cnt[i] : row[i] + p.ns[i]] = (RowT)(k - k_begin);  // slot ns = per-lane dummy
                    cnt[i] += t ? 1 : 0;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) room |= cnt[i] < p.ns[i];
        if (__ballot(room) == 0ull) break;  // every centre of this wave is full (:42)
    }

#pragma unroll
    for (int i = 0; i < NR; ++i)
        lds[p.lds_cnt_off[i] + w * PDA_WAVE + lane] = valid ? cnt[i] : 0;
    __syncthreads();

    // concatenate the S partial rows per centre, pad with the first hit, coalesced store
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int ns = p.ns[i];
        const int rs = ns + 1;
        int32_t* __restrict__ out = p.idx[i] + ((size_t)bs * p.m + tile0) * ns;
        const int nvalid = min(PDA_WAVE, p.m - tile0);
        for (int e = threadIdx.x; e < nvalid * ns; e += blockDim.x) {
            const int cc = e / ns;
            const int j = e - cc * ns;
            int off = j, val = -1, first = -1;
            for (int s = 0; s < S; ++s) {
                const int cs = lds[p.lds_cnt_off[i] + s * PDA_WAVE + cc];
                const int base = p.lds_row_off[i] + (s * PDA_WAVE + cc) * rs;
                if (first < 0 && cs > 0) first = (int)rows[base] + s * p.seglen;
                if (val < 0 && off < cs) val = (int)rows[base + off] + s * p.seglen;
                off -= cs;
            }
            if (first >= 0) out[e] = val >= 0 ? val : first;
        }
    }
}

static int launch_ball_query(const float* new_xyz, const float* xyz, int32_t* const* idx, int b,
                             int n, int m, int nr, const float* r2, float r2min,
                             const int32_t* nsamples, bool dilated, hipStream_t stream,
                             const char* what) {
    PDA_REQUIRE(b >= 0 && n >= 0 && m >= 0, "%s: negative size (b=%d n=%d m=%d)", what, b, n, m);
    PDA_REQUIRE(nr >= 1 && nr <= BQ_MAX_NR, "%s: nr=%d outside [1,%d]", what, nr, BQ_MAX_NR);
    int sum_rs = 0;
    for (int i = 0; i < nr; ++i) {
        PDA_REQUIRE(nsamples[i] >= 1, "%s: nsample[%d]=%d must be >= 1", what, i, nsamples[i]);
        PDA_REQUIRE(idx[i] != nullptr || b * m == 0, "%s: idx[%d] is null", what, i);
        sum_rs += nsamples[i] + 2;  // row (+1 pad) + count
    }
    if (b == 0 || m == 0 || n == 0) return PDA_OK;  // nothing in range: rows stay untouched
    PDA_REQUIRE(new_xyz && xyz, "%s: null input pointer", what);
    PDA_REQUIRE((int64_t)b * n * 3 < INT32_MAX && (int64_t)b * m * 64 < INT32_MAX,
                "%s: problem too large for 32-bit indexing", what);

    // S waves (= point segments) per 64-centre tile.  Pick the S that keeps the most waves in
    // flight: total waves = tiles*b*S, but a CU holds only floor(160 KiB / LDS per workgroup)
    // workgroups and 32 waves.  Segments shorter than 128 points are not worth a wave.
    // (Measured: at ONCE layer 1 -- 64 tiles x 2 scenes -- S = 2 left one wave per CU and the
    // 4096-centre query took as long as the 16384-centre one.)
    const int tiles = divup(m, PDA_WAVE);
    int S = 1;
    int64_t best_eff = 0;
    for (int cand = 1; cand <= 16; cand *= 2) {
        const int64_t rb = divup(n, cand) + BQ_BATCH <= 65535 ? 2 : 4;
        const int64_t lds = (int64_t)cand * PDA_WAVE * ((sum_rs - nr) * rb + nr * 4) + 16;
        if (lds > 150 * 1024) break;
        if (cand > 1 && divup(n, cand) < 128) break;
        const int64_t per_cu = std::min<int64_t>(160 * 1024 / lds, 32 / cand) * cand;
        const int64_t eff = std::min<int64_t>((int64_t)tiles * b * cand, 256 * per_cu);
        if (eff > best_eff) { best_eff = eff; S = cand; }
    }
    BqParams p{};
    p.new_xyz = new_xyz; p.xyz = xyz; p.n = n; p.m = m; p.r2min = r2min;
    p.seglen = divup(divup(n, S), BQ_BATCH) * BQ_BATCH;
    const bool idx16 = p.seglen <= 65535;
    const size_t row_bytes = idx16 ? 2 : 4;
    int off = 0;  // in row elements
    for (int i = 0; i < nr; ++i) {
        p.idx[i] = idx[i]; p.r2[i] = r2[i]; p.ns[i] = nsamples[i];
        p.lds_row_off[i] = off; off += S * PDA_WAVE * (nsamples[i] + 1);
    }
    const size_t rows_bytes = ((size_t)off * row_bytes + 15) / 16 * 16;
    int coff = (int)(rows_bytes / 4);  // counts: int32 units
    for (int i = 0; i < nr; ++i) { p.lds_cnt_off[i] = coff; coff += S * PDA_WAVE; }
    const size_t lds_bytes = (size_t)coff * 4;
    if (lds_bytes > 160 * 1024) {
        set_error("%s: nsample total %d needs %zu B of LDS (> 160 KiB)", what, sum_rs, lds_bytes);
        return PDA_ERR_UNSUPPORTED;
    }

    dim3 grid(tiles, b), block(S * PDA_WAVE);
    void (*kern)(const BqParams) = nullptr;
    if (idx16) {
        if (dilated) kern = ball_query_kernel<1, true, uint16_t>;
        else if (nr == 1) kern = ball_query_kernel<1, false, uint16_t>;
        else if (nr == 2) kern = ball_query_kernel<2, false, uint16_t>;
        else kern = ball_query_kernel<3, false, uint16_t>;
    } else {
        if (dilated) kern = ball_query_kernel<1, true, int32_t>;
        else if (nr == 1) kern = ball_query_kernel<1, false, int32_t>;
        else if (nr == 2) kern = ball_query_kernel<2, false, int32_t>;
        else kern = ball_query_kernel<3, false, int32_t>;
    }
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) {
            set_error("%s: cannot raise dynamic LDS to %zu B: %s", what, lds_bytes, hipGetErrorString(e));
            return PDA_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, p);
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_ball_query(const float* new_xyz, const float* xyz, int32_t* idx, int b, int n, int m,
                           float radius, int nsample, pda_stream_t stream) {
    const float r2 = radius * radius;
    int32_t* idxs[1] = {idx};
    const int32_t ns[1] = {nsample};
    return pda::launch_ball_query(new_xyz, xyz, idxs, b, n, m, 1, &r2, 0.f, ns, false,
                                  (hipStream_t)stream, "pda_ball_query");
}

PDA_API int pda_ball_query_dilated(const float* new_xyz, const float* xyz, int32_t* idx, int b, int n,
                                   int m, float max_radius, float min_radius, int nsample,
                                   pda_stream_t stream) {
    const float r1 = max_radius * max_radius;  // ball_query_gpu.cu:84
    const float r2 = min_radius * min_radius;  // :85
    int32_t* idxs[1] = {idx};
    const int32_t ns[1] = {nsample};
    return pda::launch_ball_query(new_xyz, xyz, idxs, b, n, m, 1, &r1, r2, ns, true,
                                  (hipStream_t)stream, "pda_ball_query_dilated");
}

PDA_API int pda_ball_query_multi(const float* new_xyz, const float* xyz, int32_t* const* idx, int b,
                                 int n, int m, int nr, const float* radii, const int32_t* nsamples,
                                 pda_stream_t stream) {
    if (!(nr >= 1 && nr <= pda::BQ_MAX_NR) || !radii || !nsamples || !idx) {
        pda::set_error("pda_ball_query_multi: nr=%d outside [1,%d] or null array", nr, pda::BQ_MAX_NR);
        return PDA_ERR_INVALID_ARGUMENT;
    }
    float r2[pda::BQ_MAX_NR];
    for (int i = 0; i < nr; ++i) r2[i] = radii[i] * radii[i];
    return pda::launch_ball_query(new_xyz, xyz, idx, b, n, m, nr, r2, 0.f, nsamples, false,
                                  (hipStream_t)stream, "pda_ball_query_multi");
}
