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
