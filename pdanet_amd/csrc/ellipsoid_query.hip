// ellipsoid_query.hip -- `ellipsoid_query` of the reference's extension module (pointnet2_api.cpp:16; ellipsoid_query.cpp:13-76;
// kernel ellipsoid_query_gpu.cu:311-498 with J. Burkardt's jacobi_eigenvalue, :58-298).  No PDA-SSD yaml reaches it (the
// only call on the PDA path is commented out, pointnet2_utils.py:586-587); it is built so that every name of the
// boundary has a kernel behind it.
//
// Per centre the reference runs, in ONE thread: (1) a ball query of radius e3; (2) with >= 3 hits, the covariance of the
// hits about the centre (when their mean lies >= e1/4 away) or about their mean -- skipped when a hit is exactly the
// origin, the zero matrix then stays; (3) its eigenvectors by Jacobi rotations; (4) a second pass over ALL points in the
// frame of those axes: points inside the ellipsoid (e1, e2, e3) that are not listed yet are appended until nsample.
// Here (1) IS pda_ball_query (the same rows: first nsample hits in index order, slots pre-filled with the first hit), the
// number of hits is read back from the row's padding, and one thread per centre does (2)-(4) with the point stream read
// through scalar loads (wave-uniform index).  The work arrays of the reference (ingroup_pts_cnt, ingroup_out,
// ingroup_cva, v, d: zero-filled tensors the caller never sees) live in registers.
//
// Arithmetic: the operand types of the source (float storage; the sub-expressions with double literals in double).
// Multiply-adds are not contracted (the library is built with -ffp-contract=off) except the shared squared distance of the
// first query (sqdist3): which other ones nvcc fuses cannot be read off the source, so against a CUDA build this operator's
// parity is UNPINNED (oracle/pointnet2_oracle.c holds the same statement on the CPU and the two agree bit for bit).
#include "pda_common.h"

namespace pda {

// ellipsoid_query_gpu.cu:58-298 for n = 3
__device__ void eq_jacobi3(float (&a)[9], int it_max, float (&v)[9], float (&d)[3]) {
    constexpr int n = 3;
    float bw[3], zw[3];
    for (int j = 0, k = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) v[k++] = i == j ? 1.0f : 0.0f;
    for (int i = 0; i < n; ++i) { d[i] = a[i + i * n]; bw[i] = d[i]; zw[i] = 0.0f; }
    int it_num = 0;
    while (it_num < it_max) {
        ++it_num;
        float thresh = 0.0f;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < j; ++i) thresh = thresh + a[i + j * n] * a[i + j * n];
        thresh = __fsqrt_rn(thresh) / (float)(4 * n);
        if (thresh == 0.0f) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const float gapq = (float)(10.0 * (double)fabsf(a[p + q * n]));
                const float termp = gapq + fabsf(d[p]);
                const float termq = gapq + fabsf(d[q]);
                if (4 < it_num && termp == fabsf(d[p]) && termq == fabsf(d[q])) {
                    a[p + q * n] = 0.0f;
                } else if (thresh <= fabsf(a[p + q * n])) {
                    float h = d[q] - d[p];
                    const float term = fabsf(h) + gapq;
                    float t;
                    if (term == fabsf(h)) {
                        t = a[p + q * n] / h;
                    } else {
                        const float theta = (float)(0.5 * (double)h / (double)a[p + q * n]);
                        t = (float)(1.0 / ((double)fabsf(theta) + __dsqrt_rn(1.0 + (double)(theta * theta))));
                        if (theta < 0.0f) t = -t;
                    }
                    const float c = (float)(1.0 / __dsqrt_rn(1.0 + (double)(t * t)));
                    const float s = t * c;
                    const float tau = (float)((double)s / (1.0 + (double)c));
                    h = t * a[p + q * n];
                    zw[p] = zw[p] - h; zw[q] = zw[q] + h;
                    d[p] = d[p] - h; d[q] = d[q] + h;
                    a[p + q * n] = 0.0f;
                    for (int j = 0; j < p; ++j) {
                        const float g = a[j + p * n]; h = a[j + q * n];
                        a[j + p * n] = g - s * (h + g * tau);
                        a[j + q * n] = h + s * (g - h * tau);
                    }
                    for (int j = p + 1; j < q; ++j) {
                        const float g = a[p + j * n]; h = a[j + q * n];
                        a[p + j * n] = g - s * (h + g * tau);
                        a[j + q * n] = h + s * (g - h * tau);
                    }
                    for (int j = q + 1; j < n; ++j) {
                        const float g = a[p + j * n]; h = a[q + j * n];
                        a[p + j * n] = g - s * (h + g * tau);
                        a[q + j * n] = h + s * (g - h * tau);
                    }
                    for (int j = 0; j < n; ++j) {
                        const float g = v[j + p * n]; h = v[j + q * n];
                        v[j + p * n] = g - s * (h + g * tau);
                        v[j + q * n] = h + s * (g - h * tau);
                    }
                }
            }
        for (int i = 0; i < n; ++i) { bw[i] = bw[i] + zw[i]; d[i] = bw[i]; zw[i] = 0.0f; }
    }
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < j; ++i) a[i + j * n] = a[j + i * n];
    for (int k = 0; k < n - 1; ++k) {
        int m = k;
        for (int l = k + 1; l < n; ++l)
            if (d[l] < d[m]) m = l;
        if (m != k) {
            const float t = d[m]; d[m] = d[k]; d[k] = t;
            for (int i = 0; i < n; ++i) { const float w = v[i + m * n]; v[i + m * n] = v[i + k * n]; v[i + k * n] = w; }
        }
    }
}

// idx (b, m, nsample) holds the rows of the ball query of radius e3; one thread per centre re-orients and extends its row
__global__ __launch_bounds__(64) void ellipsoid_refine_kernel(const float* __restrict__ new_xyz_all, const float* __restrict__ xyz_all,
                                                              int32_t* __restrict__ idx_all, int n, int m, float e1, float e2, float e3,
                                                              int nsample) {
    const int bs = blockIdx.y;
    const int j = blockIdx.x * 64 + threadIdx.x;
    const bool live = j < m;
    const int jj = live ? j : 0;
    const float* xyz = xyz_all + (size_t)bs * n * 3;
    const float* nw = new_xyz_all + ((size_t)bs * m + jj) * 3;
    int32_t* idx = idx_all + ((size_t)bs * m + jj) * nsample;
    const float new_x = nw[0], new_y = nw[1], new_z = nw[2];
    const float aa = e1 * e1, bb = e2 * e2, cc = e3 * e3;
    // hits of the first query: ascending distinct indices, then repeats of the first one.  (No hit and "only point 0" both
    // read as one hit: neither reaches the >= 3 below.)
    const int first = idx[0];
    int pts = 1;
    for (int s = 1; s < nsample; ++s) pts += idx[s] != first ? 1 : 0;
    const bool act = live && pts >= 3;
    float v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float sg = -1.0f;
    if (act) {
        float cva[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dd[3];
        bool flag = false;
        float means[3] = {0.0f, 0.0f, 0.0f};
        for (int k = 0; k < pts; ++k) {
            const float* pt = xyz + (size_t)idx[k] * 3;
            const float x = pt[0], y = pt[1], z = pt[2];
            flag = flag || (x == 0 && y == 0 && z == 0);
            means[0] += x; means[1] += y; means[2] += z;
        }
        if (!flag) {
            means[0] = means[0] / (float)pts; means[1] = means[1] / (float)pts; means[2] = means[2] / (float)pts;
            const float dm = __fsqrt_rn((means[0] - new_x) * (means[0] - new_x) + (means[1] - new_y) * (means[1] - new_y) +
                                        (means[2] - new_z) * (means[2] - new_z));
            const bool about_centre = (double)dm >= (double)e1 / 4.0;
            const float ox = about_centre ? new_x : means[0], oy = about_centre ? new_y : means[1], oz = about_centre ? new_z : means[2];
            float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < pts; ++k) {
                const float* pt = xyz + (size_t)idx[k] * 3;
                const float c3[3] = {pt[0] - ox, pt[1] - oy, pt[2] - oz};
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
                    for (int tn = 0; tn < 3; ++tn) acc[tn + t3 * 3] += c3[t3] * c3[tn];     // the reference's sum order: over the points
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) cva[e] = acc[e] / (float)(pts - 1);
        }
        eq_jacobi3(cva, 1000, v, dd);
        const float deter = v[6] * (v[4] * v[2] - v[1] * v[5]) - v[7] * (v[3] * v[2] - v[0] * v[5]) + v[8] * (v[3] * v[1] - v[0] * v[4]);
        sg = deter == 1.0f ? 1.0f : -1.0f;
    }
    const float r0x = sg * v[6], r0y = sg * v[7], r0z = sg * v[8];
    const float r1x = sg * v[3], r1y = sg * v[4], r1z = sg * v[5];
    const float r2x = sg * v[0], r2y = sg * v[1], r2z = sg * v[2];
    int cnt = act ? pts : nsample;                    // inactive lanes are "full" from the start
    const cfloat_ptr pts_c = as_constant(uniform_ptr(xyz));
    for (int k = 0; k < n; ++k) {
        if (__ballot(cnt < nsample) == 0ull) break;   // every lane of the wave is done
        const float s0 = pts_c[k * 3 + 0] - new_x, s1 = pts_c[k * 3 + 1] - new_y, s2 = pts_c[k * 3 + 2] - new_z;
        const float xx = r0x * s0 + r0y * s1 + r0z * s2;
        const float yy = r1x * s0 + r1y * s1 + r1z * s2;
        const float zz = r2x * s0 + r2y * s1 + r2z * s2;
        const float d3 = __fsqrt_rn((xx * xx / aa) + (yy * yy / bb) + (zz * zz / cc));
        if (cnt < nsample && d3 < 1.0f) {
            bool listed = false;
            for (int kk = 0; kk < nsample && !listed; ++kk) listed = idx[kk] == k;
            if (!listed) { idx[cnt] = k; ++cnt; }
        }
    }
}

}  // namespace pda

PDA_API int pda_ellipsoid_query(const float* new_xyz, const float* xyz, int32_t* idx, int b, int n, int m, float e1, float e2, float e3,
                                int nsample, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 1, "pda_ellipsoid_query: bad size (b=%d n=%d m=%d nsample=%d)", b, n, m, nsample);
    if (b == 0 || m == 0 || n == 0) return PDA_OK;
    PDA_REQUIRE(new_xyz && xyz && idx, "pda_ellipsoid_query: null pointer");
    PDA_REQUIRE(b <= 65535, "pda_ellipsoid_query: b = %d > 65535", b);
    // (1) the first query of the reference kernel is the ball query of radius e3 (idx zero-filled by the caller, as the
    // reference's wrapper does): same rows
    const int rc = pda_ball_query(new_xyz, xyz, idx, b, n, m, e3, nsample, stream);
    if (rc != PDA_OK) return rc;
    hipLaunchKernelGGL(pda::ellipsoid_refine_kernel, dim3(pda::divup(m, 64), b), dim3(64), 0, (hipStream_t)stream, new_xyz, xyz, idx, n, m, e1,
                       e2, e3, nsample);
    return pda::check_launch("pda_ellipsoid_query");
}
