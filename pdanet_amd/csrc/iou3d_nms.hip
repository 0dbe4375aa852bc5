// iou3d_nms.hip -- rotated BEV overlap / IoU and NMS (include/pda_train.h).
//
// Reference: iou3d_nms_kernel.cu computes an N x N/64 suppression bit mask on the GPU, copies it to the
// host (cudaMemcpy = a device synchronisation per scene and per call) and runs the greedy scan on the
// CPU (iou3d_nms.cpp:90-138).  Here the scan stays on the device: one wave per scene keeps the
// "removed" bit vector in registers (lane l owns 64-box words l, l+64, ...), walks the boxes in score
// order and ORs in the mask row of every box it keeps (one coalesced 8-byte-per-lane load); all scenes
// of a batch go through one launch pair and nothing is copied to the host.  Only the upper triangle of
// the mask is computed (the scan never reads the rest).
//
// Arithmetic: float, in the reference's order (box_overlap, iou_bev, iou_normal); cos/sin/atan2 through
// double and rounded to float, exactly as oracle/pointnet2_oracle.c does (see its comment).
#include "pda_common.h"

namespace pda {

struct Pt { float x, y; };
__device__ __forceinline__ float f_cos(float a) { return (float)cos((double)a); }
__device__ __forceinline__ float f_sin(float a) { return (float)sin((double)a); }
__device__ __forceinline__ float cross2(Pt a, Pt b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
__device__ __forceinline__ float mn(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float mx(float a, float b) { return a > b ? a : b; }
constexpr float IOU_EPS = 1e-8f;

// a box with its trigonometry done once (the reference recomputes cos/sin for every pair and corner test)
struct BevBox {
    float x, y, dx, dy, c, s;  // c, s = cos/sin(heading); cos(-h) = c, sin(-h) = -s exactly
    Pt corner[4];
};

__device__ __forceinline__ Pt rot_center(Pt ctr, float c, float s, Pt p) {
    Pt r;
    r.x = (p.x - ctr.x) * c + (p.y - ctr.y) * (-s) + ctr.x;
    r.y = (p.x - ctr.x) * s + (p.y - ctr.y) * c + ctr.y;
    return r;
}

__device__ __forceinline__ BevBox make_box(const float* b) {
    BevBox r;
    r.x = b[0]; r.y = b[1]; r.dx = b[3]; r.dy = b[4];
    r.c = f_cos(b[6]); r.s = f_sin(b[6]);
    const float hx = b[3] / 2, hy = b[4] / 2;
    const Pt ctr = {b[0], b[1]};
    const Pt raw[4] = {{b[0] - hx, b[1] - hy}, {b[0] + hx, b[1] - hy}, {b[0] + hx, b[1] + hy}, {b[0] - hx, b[1] + hy}};
#pragma unroll
    for (int k = 0; k < 4; ++k) r.corner[k] = rot_center(ctr, r.c, r.s, raw[k]);
    return r;
}

__device__ __forceinline__ bool rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {
    return mn(p1.x, p2.x) <= mx(q1.x, q2.x) && mn(q1.x, q2.x) <= mx(p1.x, p2.x) && mn(p1.y, p2.y) <= mx(q1.y, q2.y) &&
           mn(q1.y, q2.y) <= mx(p1.y, p2.y);
}

__device__ __forceinline__ bool in_box2d(const BevBox& b, Pt p) {
    const float MARGIN = 1e-2f;
    // cos(-h) = cos(h), sin(-h) = -sin(h) hold exactly for correctly rounded values
    const float c = b.c, s = -b.s;
    const float rx = (p.x - b.x) * c + (p.y - b.y) * (-s);
    const float ry = (p.x - b.x) * s + (p.y - b.y) * c;
    return fabsf(rx) < b.dx / 2 + MARGIN && fabsf(ry) < b.dy / 2 + MARGIN;
}

__device__ __forceinline__ bool seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt& ans) {
    if (!rect_cross(p0, p1, q0, q1)) return false;
    const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__device__ float box_overlap(const BevBox& a, const BevBox& b) {
    Pt cp[16];
    float ang[16];
    Pt center = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            Pt x;
            if (seg_intersection(a.corner[(i + 1) & 3], a.corner[i], b.corner[(j + 1) & 3], b.corner[j], x)) {
                cp[cnt] = x;
                center.x = center.x + x.x; center.y = center.y + x.y;
                cnt++;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(a, b.corner[k])) { center.x = center.x + b.corner[k].x; center.y = center.y + b.corner[k].y; cp[cnt++] = b.corner[k]; }
        if (in_box2d(b, a.corner[k])) { center.x = center.x + a.corner[k].x; center.y = center.y + a.corner[k].y; cp[cnt++] = a.corner[k]; }
    }
    if (cnt < 3) return 0.f;  // no polygon: the reference's area loop yields 0 (cnt - 1 < 2 terms, all from cp[0])
    center.x /= cnt; center.y /= cnt;
    for (int i = 0; i < cnt; ++i) ang[i] = (float)atan2((double)(cp[i].y - center.y), (double)(cp[i].x - center.x));
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                const Pt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
                const float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const Pt u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y}, v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
        area += cross2(u, v);
    }
    return (float)((double)fabsf(area) / 2.0);
}

__device__ __forceinline__ float iou_bev(const BevBox& a, const BevBox& b) {
    const float sa = a.dx * a.dy, sb = b.dx * b.dy, so = box_overlap(a, b);
    return so / mx(sa + sb - so, IOU_EPS);
}

__device__ __forceinline__ float iou_normal(const float* a, const float* b) {
    const float left = mx(a[0] - a[3] / 2, b[0] - b[3] / 2), right = mn(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = mx(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = mn(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = mx(right - left, 0.f), h = mx(bottom - top, 0.f);
    const float inter = w * h, sa = a[3] * a[4], sb = b[3] * b[4];
    return inter / mx(sa + sb - inter, IOU_EPS);
}

// (na,7) x (nb,7) -> (na,nb): 64 b-boxes staged per workgroup, one thread per pair
template <bool IOU>
__global__ __launch_bounds__(256) void boxes_bev_kernel(const float* __restrict__ boxes_a, const float* __restrict__ boxes_b,
                                                        float* __restrict__ out, int na, int nb) {
    __shared__ BevBox sb[64];
    const int b0 = blockIdx.x * 64, a0 = blockIdx.y * 4;
    if ((int)threadIdx.x < 64 && b0 + (int)threadIdx.x < nb) sb[threadIdx.x] = make_box(boxes_b + (size_t)(b0 + threadIdx.x) * 7);
    __syncthreads();
    const int ai = a0 + (int)(threadIdx.x >> 6), bi = b0 + (int)(threadIdx.x & 63);
    if (ai >= na || bi >= nb) return;
    const BevBox a = make_box(boxes_a + (size_t)ai * 7);
    const BevBox& b = sb[threadIdx.x & 63];
    out[(size_t)ai * nb + bi] = IOU ? iou_bev(a, b) : box_overlap(a, b);
}

// suppression masks, upper triangle: mask[scene][i][cb] bit j = IoU(box i, box cb*64+j) > thresh, j > i
template <bool NORMAL>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, unsigned long long* __restrict__ mask,
                                                      int n, int col_blocks, float thresh) {
    const int col = blockIdx.x, row = blockIdx.y, scene = blockIdx.z;
    if (col < row) return;
    boxes += (size_t)scene * n * 7;
    mask += (size_t)scene * n * col_blocks;
    __shared__ BevBox sb[64];
    __shared__ float raw[64 * 7];
    const int col_size = min(n - col * 64, 64), row_size = min(n - row * 64, 64);
    if ((int)threadIdx.x < col_size) {
        const float* p = boxes + (size_t)(col * 64 + threadIdx.x) * 7;
        if (NORMAL) {
#pragma unroll
            for (int k = 0; k < 7; ++k) raw[threadIdx.x * 7 + k] = p[k];
        } else {
            sb[threadIdx.x] = make_box(p);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x >= row_size) return;
    const int i = row * 64 + threadIdx.x;
    const float* pa = boxes + (size_t)i * 7;
    BevBox a;
    if (!NORMAL) a = make_box(pa);
    unsigned long long t = 0;
    const int start = (row == col) ? (int)threadIdx.x + 1 : 0;
    for (int j = start; j < col_size; ++j) {
        const float v = NORMAL ? iou_normal(pa, raw + j * 7) : iou_bev(a, sb[j]);
        if (v > thresh) t |= 1ULL << j;
    }
    mask[(size_t)i * col_blocks + col] = t;
}

// greedy scan, one wave per scene
constexpr int SCAN_WORDS = 8;  // 64-box words per lane: n <= 64 * 64 * 8 = 32768 boxes per scene

__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ num_valid,
                                                      long long* __restrict__ keep, int* __restrict__ num_keep, int n,
                                                      int col_blocks) {
    const int scene = blockIdx.x, lane = threadIdx.x;
    mask += (size_t)scene * n * col_blocks;
    keep += (size_t)scene * n;
    int nv = num_valid ? num_valid[scene] : n;
    nv = max(0, min(nv, n));
    unsigned long long remv[SCAN_WORDS];
#pragma unroll
    for (int w = 0; w < SCAN_WORDS; ++w) remv[w] = 0;
    int num = 0;
    for (int nb = 0; nb * 64 < nv; ++nb) {
        const int owner = nb & 63, slot = nb >> 6;
        const int in_block = min(64, nv - nb * 64);
        const unsigned long long valid = in_block == 64 ? ~0ULL : ((1ULL << in_block) - 1);
        unsigned long long mine = 0;
#pragma unroll
        for (int w = 0; w < SCAN_WORDS; ++w) if (w == slot) mine = remv[w];
        unsigned long long word = __shfl(mine, owner);
        unsigned long long avail = ~word & valid;
        while (avail) {
            const int bit = __ffsll((long long)avail) - 1;
            const int i = nb * 64 + bit;
            if (lane == 0) keep[num] = i;
            ++num;
            const unsigned long long* row = mask + (size_t)i * col_blocks;
            unsigned long long self = 0;
#pragma unroll
            for (int w = 0; w < SCAN_WORDS; ++w) {
                const int cb = w * 64 + lane;
                if (cb >= nb && cb < col_blocks) {
                    remv[w] |= row[cb];
                    if (w == slot) self = remv[w];
                }
            }
            word = __shfl(self, owner);
            const unsigned long long above = bit == 63 ? 0ULL : (~0ULL << (bit + 1));
            avail = ~word & valid & above;
        }
    }
    for (int k = num + lane; k < n; k += 64) keep[k] = -1;
    if (lane == 0) num_keep[scene] = num;
}

template <bool IOU>
static int launch_boxes_bev(const float* a, const float* b, float* out, int na, int nb, hipStream_t stream, const char* what) {
    PDA_REQUIRE(na >= 0 && nb >= 0, "%s: na=%d nb=%d", what, na, nb);
    if (na == 0 || nb == 0) return PDA_OK;
    PDA_REQUIRE(a && b && out, "%s: null pointer", what);
    PDA_REQUIRE(divup(na, 4) <= 65535, "%s: too many boxes_a (%d)", what, na);
    hipLaunchKernelGGL(boxes_bev_kernel<IOU>, dim3(divup(nb, 64), divup(na, 4)), dim3(256), 0, stream, a, b, out, na, nb);
    return check_launch(what);
}

}  // namespace pda

PDA_API int pda_boxes_overlap_bev(const float* boxes_a, const float* boxes_b, float* ans_overlap, int num_a, int num_b,
                                  pda_stream_t stream) {
    return pda::launch_boxes_bev<false>(boxes_a, boxes_b, ans_overlap, num_a, num_b, (hipStream_t)stream, "pda_boxes_overlap_bev");
}

PDA_API int pda_boxes_iou_bev(const float* boxes_a, const float* boxes_b, float* ans_iou, int num_a, int num_b,
                              pda_stream_t stream) {
    return pda::launch_boxes_bev<true>(boxes_a, boxes_b, ans_iou, num_a, num_b, (hipStream_t)stream, "pda_boxes_iou_bev");
}

PDA_API int64_t pda_nms_mask_words(int n) { return n <= 0 ? 0 : (int64_t)n * ((n + 63) / 64); }

PDA_API int pda_nms_bev(const float* boxes, const int32_t* num_valid, int64_t* keep, int32_t* num_keep, uint64_t* mask_scratch,
                        int b, int n, float thresh, int normal, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 0, "pda_nms_bev: b=%d n=%d", b, n);
    if (b == 0) return PDA_OK;
    PDA_REQUIRE(keep && num_keep, "pda_nms_bev: null output");
    PDA_REQUIRE(n <= 64 * 64 * pda::SCAN_WORDS, "pda_nms_bev: %d boxes per scene > %d", n, 64 * 64 * pda::SCAN_WORDS);
    PDA_REQUIRE(b <= 65535, "pda_nms_bev: batch %d > 65535", b);
    const int cb = pda::divup(n, 64);
    if (n > 0) {
        PDA_REQUIRE(boxes && mask_scratch, "pda_nms_bev: null pointer");
        PDA_REQUIRE(cb <= 65535, "pda_nms_bev: too many boxes");
        if (normal)
            hipLaunchKernelGGL(pda::nms_mask_kernel<true>, dim3(cb, cb, b), dim3(64), 0, (hipStream_t)stream, boxes,
                               (unsigned long long*)mask_scratch, n, cb, thresh);
        else
            hipLaunchKernelGGL(pda::nms_mask_kernel<false>, dim3(cb, cb, b), dim3(64), 0, (hipStream_t)stream, boxes,
                               (unsigned long long*)mask_scratch, n, cb, thresh);
    }
    hipLaunchKernelGGL(pda::nms_scan_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)mask_scratch,
                       num_valid, (long long*)keep, num_keep, n, cb);
    return pda::check_launch("pda_nms_bev");
}
