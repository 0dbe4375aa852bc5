// head_targets.hip -- the point-wise, gradient-free parts of the IA-SSD head's target assignment as single launches
// (include/pda_train.h).  The reference (IASSD_head.py:132-277, :889-963) and this repo's sync-free torch version of it
// build these tensors from ~20 (labels) / ~30 (soft masks) elementwise launches per point set and there are 5 + 3 point
// sets per step; the arithmetic is a handful of compares and one exp per point, so the step pays launch latency, not work.
#include "pda_common.h"

namespace pda {

// mode 0: set_ignore_flag          (:207-217)  fg = in box;            points only in the enlarged box are ignored (-1)
// mode 1: use_ex_gt_assign         (:190-205)  fg = in enlarged box;   instance points keep their own box index
// mode 2: use_ex_gt_assign + fg_pc_ignore      fg = in enlarged xor in box; index -1 for the instance points
__global__ __launch_bounds__(256) void assign_point_targets_kernel(const float* __restrict__ gt_boxes, const int* __restrict__ in_box,
                                                                   const int* __restrict__ in_ext, int64_t* __restrict__ labels,
                                                                   int64_t* __restrict__ box_idx, float* __restrict__ gt_of_points,
                                                                   int n, int t, int mode, int single_class, int64_t total) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int64_t b = p / n;
    const int ib = in_box[p], ie = in_ext[p];
    const bool box_fg = ib >= 0, ext_fg = ie >= 0;
    bool fg;
    int idx;
    int64_t lab = 0;
    if (mode == 0) {
        fg = box_fg; idx = ib;
        if (fg != ext_fg) lab = -1;
    } else {
        idx = box_fg ? ib : ie;
        if (mode == 2) { fg = ext_fg != box_fg; if (ib != -1) idx = -1; }
        else fg = ext_fg;
    }
    // gt_boxes[scene][idx]; a negative index wraps to the last row, as the reference's advanced indexing does
    const float4* row = reinterpret_cast<const float4*>(gt_boxes + ((size_t)b * t + (idx < 0 ? idx + t : idx)) * 8);
    const float4 r0 = row[0], r1 = row[1];
    const int64_t cls = single_class ? 1 : (int64_t)r1.w;
    if (fg) lab = cls;
    labels[p] = lab;
    box_idx[p] = idx;
    float4* out = reinterpret_cast<float4*>(gt_of_points + (size_t)p * 8);
    out[0] = r0; out[1] = r1;
}

// soft labels of gauss_fun_once_topk_GT_add_same_size (:889-963): exp(-0.5 |S d|^2), d = the point's offset in its box
// frame, S = diag(4/(w^2+l^2), 4/(w^2+h^2), 4/(h^2+l^2)) scaled x4 / x6 / x5 for classes 1 / 2 / 3; 0 where label <= 0
__global__ __launch_bounds__(256) void sa_gaussian_mask_kernel(const float* __restrict__ coords, int stride, int offset,
                                                               const float* __restrict__ gt, const int64_t* __restrict__ labels,
                                                               float* __restrict__ out, int64_t total) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const float* c = coords + (size_t)p * stride + offset;
    const float4 g0 = reinterpret_cast<const float4*>(gt + (size_t)p * 8)[0], g1 = reinterpret_cast<const float4*>(gt + (size_t)p * 8)[1];
    const float dx = c[0] - g0.x, dy = c[1] - g0.y, dz = c[2] - g0.z;
    const float w = g0.w, l = g1.x, h = g1.y, a = -g1.z, cls = g1.w;
    const float ca = cosf(a), sa = sinf(a);
    const float ox = dx * ca + dy * (-sa), oy = dx * sa + dy * ca, oz = dz;      // (dx, dy, dz) @ [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    const float k = cls == 1.f ? 4.f : (cls == 2.f ? 6.f : (cls == 3.f ? 5.f : 1.f));
    const float vx = ox * (4.f / (w * w + l * l) * k), vy = oy * (4.f / (w * w + h * h) * k), vz = oz * (4.f / (h * h + l * l) * k);
    const float hm = expf(-0.5f * (vx * vx + vy * vy + vz * vz));
    out[p] = labels[p] > 0 ? hm : 0.f;
}

// The whole of assign_stack_targets_IASSD (IASSD_head.py:132-277) for one point set in ONE launch: the two box queries
// (the boxes and the boxes enlarged by `extra` -- the arithmetic of points_in_boxes_kernel, csrc/points_in_boxes.hip, record by
// record), the label / index / box gather of assign_point_targets_kernel above and, when `box_labels` is given, the targets of
// PointResidual_BinOri_Coder.encode_torch (box_coder_utils.py:236-264) for the foreground points (zeros elsewhere).
struct HtBoxRec {
    float cx, cy, cz, cosa, sina, hz, hz_e;
    double lim_x, lim_y, lim_xe, lim_ye;
};

__global__ __launch_bounds__(256) void head_assign_targets_kernel(const float* __restrict__ pts, int pstride, int poff,
                                                                  const float* __restrict__ gt_boxes, float ex, float ey, float ez,
                                                                  int64_t* __restrict__ labels, int64_t* __restrict__ box_idx,
                                                                  float* __restrict__ gt_of_points, float* __restrict__ box_labels,
                                                                  const float* __restrict__ mean_size, int bins, int n, int t, int mode,
                                                                  int single_class) {
    __shared__ HtBoxRec rec[256];
    const int bs = blockIdx.y;
    const int pt = blockIdx.x * 256 + threadIdx.x;
    const bool live = pt < n;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) {
        const float* p = pts + ((size_t)bs * n + pt) * pstride + poff;
        x = p[0]; y = p[1]; z = p[2];
    }
    int ib = -1, ie = -1;
    for (int k0 = 0; k0 < t; k0 += 256) {
        const int nk = min(256, t - k0);
        __syncthreads();
        if ((int)threadIdx.x < nk) {
            const float* b = gt_boxes + ((size_t)bs * t + k0 + threadIdx.x) * 8;
            HtBoxRec r;
            r.cx = b[0]; r.cy = b[1]; r.cz = b[2];
            const double a = (double)(-b[6]);
            r.cosa = (float)cos(a);
            r.sina = (float)sin(a);
            r.hz = b[5] * 0.5f;
            r.lim_x = (double)b[3] / 2.0 + (double)1e-5f;
            r.lim_y = (double)b[4] / 2.0 + (double)1e-5f;
            const float dxe = b[3] + ex, dye = b[4] + ey, dze = b[5] + ez;       // enlarge_box3d: f32 additions
            r.hz_e = dze * 0.5f;
            r.lim_xe = (double)dxe / 2.0 + (double)1e-5f;
            r.lim_ye = (double)dye / 2.0 + (double)1e-5f;
            rec[threadIdx.x] = r;
        }
        __syncthreads();
        if (live && (ib < 0 || ie < 0)) {
            for (int k = 0; k < nk; ++k) {
                const HtBoxRec& r = rec[k];
                const float az = fabsf(z - r.cz);
                if (az > r.hz_e && az > r.hz) continue;
                const float sx = x - r.cx, sy = y - r.cy;
#if PDA_FP_CONTRACT
                const float lx = __builtin_fmaf(sx, r.cosa, sy * (-r.sina));
                const float ly = __builtin_fmaf(sx, r.sina, sy * r.cosa);
#else
                const float lx = sx * r.cosa + sy * (-r.sina);
                const float ly = sx * r.sina + sy * r.cosa;
#endif
                const double alx = (double)fabsf(lx), aly = (double)fabsf(ly);
                if (ib < 0 && !(az > r.hz) && alx < r.lim_x && aly < r.lim_y) ib = k0 + k;
                if (ie < 0 && !(az > r.hz_e) && alx < r.lim_xe && aly < r.lim_ye) ie = k0 + k;
                if (ib >= 0 && ie >= 0) break;
            }
        }
    }
    if (!live) return;
    const size_t p = (size_t)bs * n + pt;
    const bool box_fg = ib >= 0, ext_fg = ie >= 0;
    bool fg;
    int idx;
    int64_t lab = 0;
    if (mode == 0) {
        fg = box_fg; idx = ib;
        if (fg != ext_fg) lab = -1;
    } else {
        idx = box_fg ? ib : ie;
        if (mode == 2) { fg = ext_fg != box_fg; if (ib != -1) idx = -1; }
        else fg = ext_fg;
    }
    const float4* row = reinterpret_cast<const float4*>(gt_boxes + ((size_t)bs * t + (idx < 0 ? idx + t : idx)) * 8);
    const float4 r0 = row[0], r1 = row[1];
    const int64_t cls = single_class ? 1 : (int64_t)r1.w;
    if (fg) lab = cls;
    labels[p] = lab;
    box_idx[p] = idx;
    float4* out = reinterpret_cast<float4*>(gt_of_points + p * 8);
    out[0] = r0; out[1] = r1;
    if (box_labels) {
        float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;
        if (lab > 0) {
            const float PI = 3.14159265358979323846f;
            const float dxg = fmaxf(r0.w, 1e-5f), dyg = fmaxf(r1.x, 1e-5f), dzg = fmaxf(r1.y, 1e-5f);
            if (mean_size) {
                int gc = (int)r1.w; gc = gc < 1 ? 1 : gc;
                const float dxa = mean_size[(gc - 1) * 3], dya = mean_size[(gc - 1) * 3 + 1], dza = mean_size[(gc - 1) * 3 + 2];
                const float diag = sqrtf(dxa * dxa + dya * dya);
                e0 = make_float4((r0.x - x) / diag, (r0.y - y) / diag, (r0.z - z) / dza, logf(dxg / dxa));
                e1.x = logf(dyg / dya); e1.y = logf(dzg / dza);
            } else {
                e0 = make_float4(r0.x - x, r0.y - y, r0.z - z, logf(dxg));
                e1.x = logf(dyg); e1.y = logf(dzg);
            }
            const float inter = (float)(2.0 * 3.14159265358979323846 / (double)bins), half = (float)(3.14159265358979323846 / (double)bins);
            const float rg = fminf(fmaxf(r1.z, -PI + 1e-5f), PI - 1e-5f);
            const float sh = rg + PI;
            const float bin = floorf(sh / inter);
            e1.z = bin;
            e1.w = (sh - (bin * inter + half)) / half;
        }
        float4* bl = reinterpret_cast<float4*>(box_labels + p * 8);
        bl[0] = e0; bl[1] = e1;
    }
}

}  // namespace pda

PDA_API int pda_assign_point_targets(const float* gt_boxes, const int32_t* in_box, const int32_t* in_ext, int64_t* labels,
                                     int64_t* box_idx, float* gt_of_points, int b, int n, int t, int mode, int single_class,
                                     pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 0 && t >= 1 && mode >= 0 && mode <= 2, "pda_assign_point_targets: b=%d n=%d boxes=%d mode=%d", b, n, t, mode);
    const int64_t total = (int64_t)b * n;
    if (total == 0) return PDA_OK;
    PDA_REQUIRE(gt_boxes && in_box && in_ext && labels && box_idx && gt_of_points, "pda_assign_point_targets: null pointer");
    PDA_REQUIRE((((uintptr_t)gt_boxes | (uintptr_t)gt_of_points) & 15) == 0, "pda_assign_point_targets: gt_boxes / gt_of_points must be 16-byte aligned");
    hipLaunchKernelGGL(pda::assign_point_targets_kernel, dim3((unsigned)pda::divup64(total, 256)), dim3(256), 0, (hipStream_t)stream, gt_boxes,
                       in_box, in_ext, labels, box_idx, gt_of_points, n, t, mode, single_class, total);
    return pda::check_launch("pda_assign_point_targets");
}

PDA_API int pda_sa_gaussian_mask(const float* coords, int stride, int offset, const float* gt_of_points, const int64_t* labels,
                                 float* out, int64_t points, pda_stream_t stream) {
    PDA_REQUIRE(points >= 0 && stride >= 3 && offset >= 0 && offset + 3 <= stride, "pda_sa_gaussian_mask: points=%lld stride=%d offset=%d",
                (long long)points, stride, offset);
    if (points == 0) return PDA_OK;
    PDA_REQUIRE(coords && gt_of_points && labels && out, "pda_sa_gaussian_mask: null pointer");
    PDA_REQUIRE(((uintptr_t)gt_of_points & 15) == 0, "pda_sa_gaussian_mask: gt_of_points must be 16-byte aligned");
    hipLaunchKernelGGL(pda::sa_gaussian_mask_kernel, dim3((unsigned)pda::divup64(points, 256)), dim3(256), 0, (hipStream_t)stream, coords,
                       stride, offset, gt_of_points, labels, out, points);
    return pda::check_launch("pda_sa_gaussian_mask");
}

PDA_API int pda_head_assign_targets(const float* points, int point_stride, int point_offset, const float* gt_boxes, const float* extra_width,
                                    int64_t* labels, int64_t* box_idx, float* gt_of_points, float* box_labels, const float* mean_size,
                                    int bins, int b, int n, int t, int mode, int single_class, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 0 && t >= 1 && mode >= 0 && mode <= 2 && point_stride >= 3 && point_offset >= 0 && point_offset + 3 <= point_stride,
                "pda_head_assign_targets: b=%d n=%d boxes=%d mode=%d stride=%d", b, n, t, mode, point_stride);
    if (b == 0 || n == 0) return PDA_OK;
    PDA_REQUIRE(points && gt_boxes && extra_width && labels && box_idx && gt_of_points, "pda_head_assign_targets: null pointer");
    PDA_REQUIRE((((uintptr_t)gt_boxes | (uintptr_t)gt_of_points | (uintptr_t)box_labels) & 15) == 0 && b <= 65535,
                "pda_head_assign_targets: gt_boxes / gt_of_points / box_labels must be 16-byte aligned, b <= 65535");
    PDA_REQUIRE(!box_labels || (bins >= 1 && bins <= 64), "pda_head_assign_targets: bins=%d", bins);
    hipLaunchKernelGGL(pda::head_assign_targets_kernel, dim3(pda::divup(n, 256), b), dim3(256), 0, (hipStream_t)stream, points, point_stride,
                       point_offset, gt_boxes, extra_width[0], extra_width[1], extra_width[2], labels, box_idx, gt_of_points, box_labels,
                       mean_size, bins, n, t, mode, single_class);
    return pda::check_launch("pda_head_assign_targets");
}
