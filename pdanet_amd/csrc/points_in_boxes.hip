// points_in_boxes.hip -- pda_points_in_boxes (include/pda_train.h), the target-assignment kernel of
// the IA-SSD head (IASSD_head.py:132-277 calls it up to 6x per scene and step).
//
// Reference: one thread per point looping over the boxes, cosf/sinf of the heading recomputed for
// every (point, box) pair (roiaware_pool3d_kernel.cu:16-36,313-336).  Here a workgroup first turns up
// to 256 boxes into LDS records (centre, cos, sin, half-extent limits) -- the trigonometry is done
// once per box, in double and rounded to float (see oracle/pointnet2_oracle.c on why) -- then every
// lane tests its point against the records in ascending box order; LDS reads are wave-uniform
// (broadcast).  HBM traffic is the compulsory pts + boxes + idx; the kernel is latency/VALU-trivial.
#include "pda_common.h"

namespace pda {

struct BoxRec {
    float cx, cy, cz, cosa, sina, hz;  // hz = dz/2 (exact in float)
    double lim_x, lim_y;               // dx/2.0 + MARGIN, dy/2.0 + MARGIN as the reference's double expression
};

__global__ __launch_bounds__(256) void points_in_boxes_kernel(const float* __restrict__ boxes, const float* __restrict__ pts,
                                                              int* __restrict__ out, int t, int m) {
    __shared__ BoxRec rec[256];
    const int bs = blockIdx.y;
    const int pt = blockIdx.x * 256 + threadIdx.x;
    const bool live = pt < m;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) {
        const float* p = pts + ((size_t)bs * m + pt) * 3;
        x = p[0]; y = p[1]; z = p[2];
    }
    int found = -1;
    for (int k0 = 0; k0 < t; k0 += 256) {
        const int nk = min(256, t - k0);
        __syncthreads();
        if ((int)threadIdx.x < nk) {
            const float* b = boxes + ((size_t)bs * t + k0 + threadIdx.x) * 7;
            BoxRec r;
            r.cx = b[0]; r.cy = b[1]; r.cz = b[2];
            r.hz = b[5] * 0.5f;
            const double a = (double)(-b[6]);
            r.cosa = (float)cos(a);
            r.sina = (float)sin(a);
            r.lim_x = (double)b[3] / 2.0 + (double)1e-5f;
            r.lim_y = (double)b[4] / 2.0 + (double)1e-5f;
            rec[threadIdx.x] = r;
        }
        __syncthreads();
        if (live && found < 0) {
            for (int k = 0; k < nk; ++k) {
                const BoxRec& r = rec[k];
                if (fabsf(z - r.cz) > r.hz) continue;
                const float sx = x - r.cx, sy = y - r.cy;
#if PDA_FP_CONTRACT
                const float lx = __builtin_fmaf(sx, r.cosa, sy * (-r.sina));
                const float ly = __builtin_fmaf(sx, r.sina, sy * r.cosa);
#else
                const float lx = sx * r.cosa + sy * (-r.sina);
                const float ly = sx * r.sina + sy * r.cosa;
#endif
                if ((int)((double)fabsf(lx) < r.lim_x) & (int)((double)fabsf(ly) < r.lim_y)) {
                    found = k0 + k;
                    break;
                }
            }
        }
    }
    if (live && found >= 0) out[(size_t)bs * m + pt] = found;
}

}  // namespace pda

PDA_API int pda_points_in_boxes(const float* boxes, const float* pts, int32_t* box_idx_of_points, int b, int t, int m,
                                pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && t >= 0 && m >= 0, "pda_points_in_boxes: b=%d t=%d m=%d", b, t, m);
    if (b == 0 || m == 0 || t == 0) return PDA_OK;
    PDA_REQUIRE(boxes && pts && box_idx_of_points, "pda_points_in_boxes: null pointer");
    PDA_REQUIRE(b <= 65535, "pda_points_in_boxes: batch %d > 65535", b);
    hipLaunchKernelGGL(pda::points_in_boxes_kernel, dim3(pda::divup(m, 256), b), dim3(256), 0, (hipStream_t)stream, boxes,
                       pts, box_idx_of_points, t, m);
    return pda::check_launch("pda_points_in_boxes");
}
