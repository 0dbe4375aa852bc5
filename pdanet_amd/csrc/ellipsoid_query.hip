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

// Eigen-decomposition of a symmetric 3x3 matrix by cyclic Jacobi rotations: the sweep order (0,1), (0,2), (1,2), the
// threshold, the small-element rule after four sweeps, the rotation formulas and the final ascending sort are those of
// the routine the reference calls (ellipsoid_query_gpu.cu:58-298, n = 3), so that the axes -- and with them the second
// query -- come out the same; written for the three off-diagonal elements as scalars instead of an n x n array.
// On return: evec[3 k + i] = component i of the eigenvector of the k-th smallest eigenvalue.
struct Sym3Jacobi {
    float diag[3];          // running diagonal
    float off[3];           // upper triangle: off[0] = (0,1), off[1] = (0,2), off[2] = (1,2)
    float evec[9];          // column-major eigenvector estimate
    float base[3], corr[3]; // diagonal at the start of the sweep and the corrections gathered during it

    // the plane rotation applied to a pair (g, h): g' = g - s (h + g tau), h' = h + s (g - h tau)
    static __device__ __forceinline__ void spin(float& g, float& h, float s, float tau) {
        const float g0 = g, h0 = h;
        g = g0 - s * (h0 + g0 * tau);
        h = h0 + s * (g0 - h0 * tau);
    }

    // one (p, q) step of a sweep; `e` is the off-diagonal element (p, q), (x, y) the two other off-diagonal elements in
    // the order the rotation pairs them
    __device__ __forceinline__ void step(int p, int q, float& e, float& x, float& y, bool late, float floor_) {
        const float mag = fabsf(e);
        const float guard = (float)(10.0 * (double)mag);
        if (late && guard + fabsf(diag[p]) == fabsf(diag[p]) && guard + fabsf(diag[q]) == fabsf(diag[q])) {
            e = 0.0f;                                   // too small to change either eigenvalue any more
            return;
        }
        if (!(floor_ <= mag)) return;
        const float gap = diag[q] - diag[p];
        float t;
        if (fabsf(gap) + guard == fabsf(gap)) {
            t = e / gap;
        } else {
            const float theta = (float)(0.5 * (double)gap / (double)e);
            t = (float)(1.0 / ((double)fabsf(theta) + __dsqrt_rn(1.0 + (double)(theta * theta))));
            t = theta < 0.0f ? -t : t;
        }
        const float c = (float)(1.0 / __dsqrt_rn(1.0 + (double)(t * t)));
        const float s = t * c;
        const float tau = (float)((double)s / (1.0 + (double)c));
        const float shift = t * e;
        corr[p] = corr[p] - shift; corr[q] = corr[q] + shift;
        diag[p] = diag[p] - shift; diag[q] = diag[q] + shift;
        e = 0.0f;
        spin(x, y, s, tau);
#pragma unroll
        for (int i = 0; i < 3; ++i) spin(evec[3 * p + i], evec[3 * q + i], s, tau);
    }

    __device__ void run(const float (&m)[9], int max_sweeps) {
#pragma unroll
        for (int k = 0; k < 9; ++k) evec[k] = (k % 4 == 0) ? 1.0f : 0.0f;
        diag[0] = m[0]; diag[1] = m[4]; diag[2] = m[8];
        off[0] = m[3]; off[1] = m[6]; off[2] = m[7];            // a[i + 3 j] with i < j
#pragma unroll
        for (int i = 0; i < 3; ++i) { base[i] = diag[i]; corr[i] = 0.0f; }
        for (int sweep = 1; sweep <= max_sweeps; ++sweep) {
            float ss = 0.0f;
            ss = ss + off[0] * off[0]; ss = ss + off[1] * off[1]; ss = ss + off[2] * off[2];
            const float floor_ = __fsqrt_rn(ss) / 12.0f;
            if (floor_ == 0.0f) break;
            const bool late = sweep > 4;
            step(0, 1, off[0], off[1], off[2], late, floor_);   // (0,1) pairs (0,2) with (1,2)
            step(0, 2, off[1], off[0], off[2], late, floor_);   // (0,2) pairs (0,1) with (1,2)
            step(1, 2, off[2], off[0], off[1], late, floor_);   // (1,2) pairs (0,1) with (0,2)
#pragma unroll
            for (int i = 0; i < 3; ++i) { base[i] = base[i] + corr[i]; diag[i] = base[i]; corr[i] = 0.0f; }
        }
        // ascending eigenvalues (selection sort, as the reference orders them), eigenvector columns follow
        for (int k = 0; k < 2; ++k) {
            int lo = k;
            for (int l = k + 1; l < 3; ++l) lo = diag[l] < diag[lo] ? l : lo;
            if (lo != k) {
                const float d0 = diag[lo]; diag[lo] = diag[k]; diag[k] = d0;
#pragma unroll
                for (int i = 0; i < 3; ++i) { const float w = evec[3 * lo + i]; evec[3 * lo + i] = evec[3 * k + i]; evec[3 * k + i] = w; }
            }
        }
    }
};

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
        float cva[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
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
        Sym3Jacobi jac;
        jac.run(cva, 1000);
#pragma unroll
        for (int e = 0; e < 9; ++e) v[e] = jac.evec[e];
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
