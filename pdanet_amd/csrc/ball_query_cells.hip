// ball_query_cells.hip -- ball query through a uniform cell list, index-exact w.r.t. the reference kernel
// ball_query_kernel_fast (/root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:9-45):
// "the first nsample points, in ascending index, with d2 < r*r; the first hit pre-fills the row; no hit: row untouched".
//
// csrc/ball_query.hip scans all N points for every centre (M*N distance tests; VALU-bound at 0.66 of the lane-op
// peak).  The answer of a centre only depends on the points inside its ball, so here (N >= 8192):
//   1. bin the points of a scene into cells of edge g >= 1.01 * r_max (bounding box, counting sort by atomics; the
//      order inside a cell is arbitrary -- step 3 sorts by index anyway);
//   2. one WAVE per centre: its 27 neighbouring cells are 9 contiguous runs of the sorted records {x, y, z, index};
//      64 candidates per step, the same float expression and operand order as the reference (`sqdist3(new, x)`), hits
//      appended (ballot + prefix count) to a per-radius list in LDS;
//   3. the nsample SMALLEST indices of the list are selected by bisection on the index value (counting with ballots),
//      ranked and written in ascending order; the rest of the row repeats the first one;
//   4. a centre whose list overflows (more than LCAP hits: a dense neighbourhood) falls back to the reference's own
//      scan, 64 indices per step in ascending order with early exit -- in a dense ball the first nsample hits come
//      within the first nsample / hits * N indices, so that scan is short exactly where the list is long.
// Up to 3 radii share the candidate pass.  The set of hits is the brute-force set (same expression, and the 27 cells
// cover the ball: |dx| < r <= g / 1.01 moves the cell coordinate by less than one even with rounding), hence the rows
// are bit-identical.  Scratch comes from the caller (pda_ball_query_cells_scratch_bytes); nothing is allocated here.
#include "pda_common.h"

namespace pda {

constexpr int BQC_MAX_NR = 3;
constexpr int BQC_LCAP = 512;          // hits kept per (centre, radius) before the index-order fallback
constexpr int BQC_MAX_CELLS = 1 << 17;  // cells per scene
constexpr int BQC_WAVES = 4;

struct BqcGrid {        // per scene, written by bqc_grid_kernel
    float ox, oy, oz, inv_gx, inv_gy, inv_gz;
    int nx, ny, nz, ncell;
};

__device__ __forceinline__ int bqc_coord(float v, float o, float inv_g, int dim) {
    const int c = (int)floorf((v - o) * inv_g);
    return min(max(c, -1), dim);      // -1 / dim: outside the grid (centres only; points are inside by construction)
}

// one workgroup per scene: bounding box of the points -> grid
__global__ __launch_bounds__(1024) void bqc_grid_kernel(const float* __restrict__ xyz_all, BqcGrid* __restrict__ grids, int n, float g_min,
                                                        int max_cells) {
    __shared__ float red[16 * 6];
    const float* xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    const float INF = __builtin_inff();
    float lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
    for (int k = threadIdx.x; k < n; k += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = xyz[k * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    const int w = wave_id(), lane = lane_id();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float mn = -wave_max_f32(-lo[a]), mx = wave_max_f32(hi[a]);
        if (lane == 0) { red[w * 6 + a] = mn; red[w * 6 + 3 + a] = mx; }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float mn = INF, mx = -INF;
        for (int q = 0; q < 16; ++q) { mn = fminf(mn, red[q * 6 + a]); mx = fmaxf(mx, red[q * 6 + 3 + a]); }
        lo[a] = mn; hi[a] = mx;
    }
    // Cell edges >= g_min per axis, at most `max_cells` cells (the table is scanned by one workgroup per scene, so a small
    // table is worth more than thin cells).  LiDAR scenes are flat: the vertical axis is coarsened first (doubling its
    // edge costs a few extra candidates per centre, the horizontal edges cost their square), then all three grow.
    float g[3] = {g_min, g_min, g_min};
    int nc[3];
    const float ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    const int thin = (ext[2] <= ext[0] && ext[2] <= ext[1]) ? 2 : (ext[1] <= ext[0] ? 1 : 0);
    // A scene with a non-finite coordinate has no finite bounding box (ext = inf or NaN keeps nc at its cap however far the
    // edges grow): ONE cell for the whole scene, i.e. every centre tests every point -- the reference's scan, in which a
    // non-finite point is never a hit (inf < r2 and NaN < r2 are false).  Bit test: the build sets -fno-honor-nans.
    bool finite = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) finite = finite && (__float_as_uint(ext[a]) & 0x7f800000u) != 0x7f800000u;
    bool fits = false;
    for (int pass = 0; finite && pass < 400; ++pass) {   // every pass grows an edge by >= 26 %: 400 passes cover any float extent
#pragma unroll
        for (int a = 0; a < 3; ++a) nc[a] = (int)fminf(floorf(ext[a] / g[a]), 32766.f) + 1;
        if ((int64_t)nc[0] * nc[1] * nc[2] <= max_cells) { fits = true; break; }
        if (nc[thin] > 1) g[thin] *= 2.0f;
        else { g[0] *= 1.26f; g[1] *= 1.26f; g[2] *= 1.26f; }
    }
    if (!fits) {
        BqcGrid one;
        one.ox = one.oy = one.oz = 0.f;
        one.inv_gx = one.inv_gy = one.inv_gz = 0.f;      // every finite coordinate maps to cell 0
        one.nx = one.ny = one.nz = one.ncell = 1;
        grids[blockIdx.x] = one;
        return;
    }
    BqcGrid gr;
    gr.ox = lo[0]; gr.oy = lo[1]; gr.oz = lo[2];
    gr.inv_gx = 1.0f / g[0]; gr.inv_gy = 1.0f / g[1]; gr.inv_gz = 1.0f / g[2];
    gr.nx = nc[0]; gr.ny = nc[1]; gr.nz = nc[2]; gr.ncell = nc[0] * nc[1] * nc[2];
    grids[blockIdx.x] = gr;
}

__device__ __forceinline__ int bqc_cell_of_point(const BqcGrid& gr, float x, float y, float z) {
    const int ix = min(max(bqc_coord(x, gr.ox, gr.inv_gx, gr.nx), 0), gr.nx - 1);
    const int iy = min(max(bqc_coord(y, gr.oy, gr.inv_gy, gr.ny), 0), gr.ny - 1);
    const int iz = min(max(bqc_coord(z, gr.oz, gr.inv_gz, gr.nz), 0), gr.nz - 1);
    return (iz * gr.ny + iy) * gr.nx + ix;   // x fastest: the 3 x-neighbours of a cell are contiguous
}

// Lanes of a wave that fall into the same cell add ONCE: points arrive shuffled, and a LiDAR-like scene puts thousands
// of them into a handful of near-range cells -- one atomic per point serialises on those addresses (measured 67 + 93 us
// for the two passes at 2 x 65536 points).  Returns the lane's slot: the leader's fetch-add result + rank among equals.
__device__ __forceinline__ int bqc_wave_slot(int32_t* table, int cell, bool active) {
    unsigned long long rem = __ballot(active);
    int slot = 0;
    while (rem != 0ull) {
        const int leader = (int)__builtin_ctzll(rem);
        const int lc = __builtin_amdgcn_readlane(cell, leader);
        const unsigned long long eq = __ballot(active && cell == lc) & rem;
        int base = 0;
        if (lane_id() == leader) base = atomicAdd(table + lc, (int)__builtin_popcountll(eq));
        base = __builtin_amdgcn_readlane(base, leader);
        if (active && cell == lc) slot = base + (int)__builtin_popcountll(eq & ((1ull << lane_id()) - 1ull));
        rem &= ~eq;
    }
    return slot;
}

// counts[scene][cell] += 1 (counts zero-filled by the launcher)
__global__ __launch_bounds__(256) void bqc_count_kernel(const float* __restrict__ xyz_all, const BqcGrid* __restrict__ grids,
                                                        int32_t* __restrict__ counts, int n, int tstride) {
    const int s = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const bool active = k < n;
    const BqcGrid gr = grids[s];
    const float* p = xyz_all + ((size_t)s * n + (active ? k : 0)) * 3;
    bqc_wave_slot(counts + (size_t)s * tstride, bqc_cell_of_point(gr, p[0], p[1], p[2]), active);
}

// one workgroup per scene: exclusive scan of the cell counts in place -> starts; cursor = copy of the starts.
// Tiles of 16384 cells go through LDS (coalesced both ways); thread t scans its 16 consecutive cells of the tile.
__global__ __launch_bounds__(1024) void bqc_scan_kernel(const BqcGrid* __restrict__ grids, int32_t* __restrict__ counts,
                                                        int32_t* __restrict__ cursor, int tstride) {
    constexpr int PER = 16, TILE = 1024 * PER;
    __shared__ int32_t tile[TILE + TILE / 16];     // 16-int rows padded by one: conflict-free row scans
    __shared__ int32_t part[1024];
    const int s = blockIdx.x, t = threadIdx.x;
    const int ncell = grids[s].ncell;
    int32_t* c = counts + (size_t)s * tstride;
    int32_t* cur = cursor + (size_t)s * tstride;
    int32_t carry = 0;
    for (int base = 0; base < ncell; base += TILE) {
        for (int i = t; i < TILE; i += 1024) tile[i + i / 16] = base + i < ncell ? c[base + i] : 0;
        __syncthreads();
        int32_t sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) sum += tile[t * (PER + 1) + i];
        part[t] = sum;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int32_t v = t >= o ? part[t - o] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        int32_t run = carry + part[t] - sum;
#pragma unroll
        for (int i = 0; i < PER; ++i) { const int32_t v = tile[t * (PER + 1) + i]; tile[t * (PER + 1) + i] = run; run += v; }
        const int32_t total = part[1023];
        __syncthreads();
        for (int i = t; i < TILE; i += 1024)
            if (base + i < ncell) { const int32_t v = tile[i + i / 16]; c[base + i] = v; cur[base + i] = v; }
        __syncthreads();
        carry += total;
    }
    if (t == 0) c[ncell] = carry;
}

// records[scene][cursor[cell]++] = {x, y, z, index}
__global__ __launch_bounds__(256) void bqc_scatter_kernel(const float* __restrict__ xyz_all, const BqcGrid* __restrict__ grids,
                                                          int32_t* __restrict__ cursor, float4* __restrict__ records, int n,
                                                          int tstride) {
    const int s = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const bool active = k < n;
    const BqcGrid gr = grids[s];
    const float* p = xyz_all + ((size_t)s * n + (active ? k : 0)) * 3;
    const float x = p[0], y = p[1], z = p[2];
    const int slot = bqc_wave_slot(cursor + (size_t)s * tstride, bqc_cell_of_point(gr, x, y, z), active);
    if (active) records[(size_t)s * n + slot] = make_float4(x, y, z, __int_as_float(k));
}

struct BqcParams {
    const float* new_xyz;
    const float* xyz;
    const BqcGrid* grids;
    const int32_t* starts;
    const float4* records;
    int32_t* idx[BQC_MAX_NR];
    float r2[BQC_MAX_NR];
    int ns[BQC_MAX_NR];
    int n, m, tstride;
};

// number of set bits of `mask` below my lane
__device__ __forceinline__ int lane_prefix(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// The `want` smallest values of list[0, len) (distinct, non-negative), ascending, into row[0, want); len <= BQC_LCAP.
// One wave; `sel` is wave-private LDS scratch of >= 128 ints.
// Returns the smallest value (wave-uniform).
__device__ __forceinline__ int bqc_select_sorted(const int32_t* list, int len, int want, int32_t* sel, int32_t* __restrict__ row,
                                                 int row_off, int lane, int n) {
    constexpr int Q = BQC_LCAP / 64;
    int32_t v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = (lane + 64 * q) < len ? list[lane + 64 * q] : 0x7fffffff;
    int thr = 0x7fffffff;               // keep values < thr
    if (len > want) {                    // smallest t with count(v < t) >= want, by bisection on [0, n]
        int lo = 0, hi = n;              // count(v < lo) < want <= count(v < hi)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < Q; ++q) cnt += __builtin_popcountll(__ballot(v[q] < mid));
            if (cnt >= want) hi = mid; else lo = mid;
        }
        thr = hi;                        // values are distinct: exactly `want` of them are < hi
    }
    int k = 0;
    uint32_t vmin = 0x7fffffffu;
#pragma unroll
    for (int q = 0; q < Q; ++q) vmin = min(vmin, (uint32_t)v[q]);
    const int first = (int)wave_min_u32(vmin);
#pragma unroll
    for (int q = 0; q < Q; ++q) {        // compact the kept values (at most 128 = max nsample)
        const bool keep = v[q] < thr;
        const unsigned long long mk = __ballot(keep);
        if (keep) sel[k + lane_prefix(mk)] = v[q];
        k += __builtin_popcountll(mk);
    }
    // rank sort: element e goes to position #{kept values smaller than e}
    for (int e = lane; e < k; e += 64) {
        const int mine = sel[e];
        int rank = 0;
        for (int i = 0; i < k; ++i) rank += sel[i] < mine ? 1 : 0;
        row[row_off + rank] = mine;
    }
    return first;
}

template <int NR>
__global__ __launch_bounds__(BQC_WAVES * 64) void ball_query_cells_kernel(const BqcParams p) {
    __shared__ int32_t lists[BQC_WAVES][NR][BQC_LCAP];
    __shared__ int32_t selbuf[BQC_WAVES][128];
    const int w = wave_id(), lane = lane_id();
    const int s = blockIdx.y;
    const int c = blockIdx.x * BQC_WAVES + w;
    if (c >= p.m) return;                                       // wave-uniform
    const BqcGrid gr = p.grids[s];
    const float* q = p.new_xyz + ((size_t)s * p.m + c) * 3;
    const float cx = q[0], cy = q[1], cz = q[2];
    const int32_t* starts = p.starts + (size_t)s * p.tstride;
    const float4* rec = p.records + (size_t)s * p.n;
    const int ix = bqc_coord(cx, gr.ox, gr.inv_gx, gr.nx), iy = bqc_coord(cy, gr.oy, gr.inv_gy, gr.ny), iz = bqc_coord(cz, gr.oz, gr.inv_gz, gr.nz);
    const int x0 = max(ix - 1, 0), x1 = min(ix + 1, gr.nx - 1);
    int len[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) len[i] = 0;
    if (x0 <= x1) {
        for (int dz = -1; dz <= 1; ++dz) {
            const int zz = iz + dz;
            if (zz < 0 || zz >= gr.nz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = iy + dy;
                if (yy < 0 || yy >= gr.ny) continue;
                const int base = (zz * gr.ny + yy) * gr.nx;
                const int k_begin = starts[base + x0], k_end = starts[base + x1 + 1];
                for (int k0 = k_begin; k0 < k_end; k0 += 64) {
                    const int k = k0 + lane;
                    const bool in = k < k_end;
                    const float4 r = rec[in ? k : k_begin];
                    const float d2 = sqdist3(cx, cy, cz, r.x, r.y, r.z);        // (new - x), ball_query_gpu.cu:33
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const bool hit = in && d2 < p.r2[i];
                        const unsigned long long mk = __ballot(hit);
                        const int pos = len[i] + lane_prefix(mk);
                        if (hit && pos < BQC_LCAP) lists[w][i][pos] = __float_as_int(r.w);
                        len[i] += __builtin_popcountll(mk);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int ns = p.ns[i];
        int32_t* row = p.idx[i] + ((size_t)s * p.m + c) * ns;
        if (len[i] == 0) continue;                                  // no hit: the row stays untouched
        int k, first = 0;
        if (len[i] <= BQC_LCAP) {
            k = min(len[i], ns);
            first = bqc_select_sorted(lists[w][i], len[i], k, selbuf[w], row, 0, lane, p.n);
        } else {
            // dense ball: the reference's scan, 64 indices per step, ascending, early exit at nsample hits
            const float* pts = p.xyz + (size_t)s * p.n * 3;
            k = 0;
            for (int k0 = 0; k0 < p.n && k < ns; k0 += 64) {
                const int kk = k0 + lane;
                const bool in = kk < p.n;
                const int ks = in ? kk : p.n - 1;
                const float d2 = sqdist3(cx, cy, cz, pts[ks * 3 + 0], pts[ks * 3 + 1], pts[ks * 3 + 2]);
                const bool hit = in && d2 < p.r2[i];
                const unsigned long long mk = __ballot(hit);
                const int pos = k + lane_prefix(mk);
                if (hit && pos < ns) row[pos] = kk;
                if (k == 0 && mk != 0ull) first = k0 + (int)__builtin_ctzll(mk);
                k = min(ns, k + (int)__builtin_popcountll(mk));
            }
        }
        // the rest of the row repeats the first hit (ball_query_gpu.cu:35-39)
        for (int j = k + lane; j < ns; j += 64) row[j] = first;
    }
}

}  // namespace pda

PDA_API int64_t pda_ball_query_cells_scratch_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    const int64_t grids = ((int64_t)b * sizeof(pda::BqcGrid) + 255) / 256 * 256;
    const int64_t table = (int64_t)b * (pda::BQC_MAX_CELLS + 1) * 4;
    return grids + 2 * ((table + 255) / 256 * 256) + (int64_t)b * n * 16 + 256;
}

PDA_API int pda_ball_query_cells(const float* new_xyz, const float* xyz, int32_t* const* idx, int b, int n, int m, int nr,
                                 const float* radii, const int32_t* nsamples, void* scratch, int64_t scratch_bytes,
                                 pda_stream_t stream) {
    using namespace pda;
    PDA_REQUIRE(b >= 0 && n >= 0 && m >= 0, "pda_ball_query_cells: negative size (b=%d n=%d m=%d)", b, n, m);
    PDA_REQUIRE(nr >= 1 && nr <= BQC_MAX_NR && radii && nsamples && idx, "pda_ball_query_cells: nr=%d outside [1,%d] or null array", nr, BQC_MAX_NR);
    float rmax = 0.f;
    for (int i = 0; i < nr; ++i) {
        PDA_REQUIRE(nsamples[i] >= 1 && nsamples[i] <= 128, "pda_ball_query_cells: nsample[%d]=%d outside [1,128]", i, nsamples[i]);
        PDA_REQUIRE(idx[i] != nullptr || (int64_t)b * m == 0, "pda_ball_query_cells: idx[%d] is null", i);
        PDA_REQUIRE(radii[i] > 0.f && radii[i] < 1e18f, "pda_ball_query_cells: radius[%d]=%g", i, (double)radii[i]);
        rmax = radii[i] > rmax ? radii[i] : rmax;
    }
    if (b == 0 || m == 0 || n == 0) return PDA_OK;
    PDA_REQUIRE(new_xyz && xyz && scratch, "pda_ball_query_cells: null pointer");
    PDA_REQUIRE(b <= 65535 && (int64_t)b * n * 3 < INT32_MAX && (int64_t)b * m * 128 < INT32_MAX, "pda_ball_query_cells: problem too large");
    PDA_REQUIRE(scratch_bytes >= pda_ball_query_cells_scratch_bytes(b, n) && ((uintptr_t)scratch & 255) == 0,
                "pda_ball_query_cells: scratch of %lld bytes (256-byte aligned) needed", (long long)pda_ball_query_cells_scratch_bytes(b, n));
    const hipStream_t st = (hipStream_t)stream;
    char* base = (char*)scratch;
    BqcGrid* grids = (BqcGrid*)base;
    const int64_t grids_b = ((int64_t)b * sizeof(BqcGrid) + 255) / 256 * 256;
    const int64_t table_b = ((int64_t)b * (BQC_MAX_CELLS + 1) * 4 + 255) / 256 * 256;
    int32_t* counts = (int32_t*)(base + grids_b);
    int32_t* cursor = (int32_t*)(base + grids_b + table_b);
    float4* records = (float4*)(base + grids_b + 2 * table_b);
    // table size: about one cell per point, a power of two in [4096, BQC_MAX_CELLS]
    int max_cells = 4096;
    while (max_cells < n && max_cells < BQC_MAX_CELLS) max_cells *= 2;
    const int tstride = max_cells + 1;
    if (hipMemsetAsync(counts, 0, (size_t)b * tstride * 4, st) != hipSuccess) {
        set_error("pda_ball_query_cells: hipMemsetAsync failed");
        return PDA_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(bqc_grid_kernel, dim3(b), dim3(1024), 0, st, xyz, grids, n, rmax * 1.01f, max_cells);
    hipLaunchKernelGGL(bqc_count_kernel, dim3(divup(n, 256), b), dim3(256), 0, st, xyz, grids, counts, n, tstride);
    hipLaunchKernelGGL(bqc_scan_kernel, dim3(b), dim3(1024), 0, st, grids, counts, cursor, tstride);
    hipLaunchKernelGGL(bqc_scatter_kernel, dim3(divup(n, 256), b), dim3(256), 0, st, xyz, grids, cursor, records, n, tstride);
    BqcParams p{};
    p.new_xyz = new_xyz; p.xyz = xyz; p.grids = grids; p.starts = counts; p.records = records; p.n = n; p.m = m; p.tstride = tstride;
    for (int i = 0; i < nr; ++i) { p.idx[i] = idx[i]; p.r2[i] = radii[i] * radii[i]; p.ns[i] = nsamples[i]; }
    const dim3 grid(divup(m, BQC_WAVES), b), block(BQC_WAVES * 64);
    if (nr == 1) hipLaunchKernelGGL(ball_query_cells_kernel<1>, grid, block, 0, st, p);
    else if (nr == 2) hipLaunchKernelGGL(ball_query_cells_kernel<2>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(ball_query_cells_kernel<3>, grid, block, 0, st, p);
    return check_launch("pda_ball_query_cells");
}
