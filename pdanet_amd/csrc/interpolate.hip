// interpolate.hip -- three_nn / three_interpolate (+grad) for gfx950.
// Reference kernels: three_nn_kernel_fast (interpolate_gpu.cu:16-59),
// three_interpolate_kernel_fast (:84-104), three_interpolate_grad_kernel_fast (:127-149)
// under /root/reference/pcdet/ops/pointnet2/pointnet2_batch/src/.
//
// three_nn: one lane per unknown point; the known point under test is wave-uniform and is
// fetched with scalar loads (same scheme as ball_query.hip), so the kernel is pure VALU.
// The reference keeps best1..3 in double initialised to 1e40 and inserts with strict '<';
// every value it ever stores is a float, so float bests initialised to +inf give identical
// comparisons and identical outputs ((float)1e40 == +inf for the m < 3 case, :57).
#include "pda_common.h"

namespace pda {

constexpr int NN_BATCH = 8;

// 64 unknown points per workgroup (one per lane); the S waves of the workgroup scan S contiguous
// segments of the known points and the S partial top-3 lists are merged in LDS by the
// lexicographic key (d2, index) -- exactly the set and order the reference's sequential strict-'<'
// scan produces -- so a small query (n/64 tiles x B) still fills the chip.
__global__ __launch_bounds__(512) void three_nn_kernel(const float* __restrict__ unknown,
                                                        const float* __restrict__ known,
                                                        float* __restrict__ dist2,
                                                        int32_t* __restrict__ idx, int n, int m, int seglen) {
    __shared__ float sd[8][3][64];
    __shared__ int si[8][3][64];
    const int bs = blockIdx.y;
    const int w = wave_id(), lane = lane_id(), S = (int)(blockDim.x >> 6);
    const int pt = blockIdx.x * 64 + lane;
    const bool valid = pt < n;
    const float* uu = unknown + ((size_t)bs * n + min(pt, n - 1)) * 3;
    const float ux = uu[0], uy = uu[1], uz = uu[2];
    const cfloat_ptr kn = as_constant(uniform_ptr(known + (size_t)bs * m * 3));
    float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
    int i1 = 0, i2 = 0, i3 = 0;
    const int k_begin = w * seglen, k_end = min(m, k_begin + seglen);
    for (int k0 = k_begin; k0 < k_end; k0 += NN_BATCH) {
        float px[NN_BATCH], py[NN_BATCH], pz[NN_BATCH];
        const bool full = k0 + NN_BATCH <= k_end;
#pragma unroll
        for (int u = 0; u < NN_BATCH; ++u) {
            const int k = full ? k0 + u : min(k0 + u, m - 1);
            px[u] = kn[k * 3 + 0]; py[u] = kn[k * 3 + 1]; pz[u] = kn[k * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < NN_BATCH; ++u) {
            const int k = k0 + u;
            float d = sqdist3(ux, uy, uz, px[u], py[u], pz[u]);  // (u - x), interpolate_gpu.cu:43
            if (!full && k >= k_end) d = __builtin_inff();        // outside my segment: never '<'
            // branch-free form of the if / else-if / else-if chain (:44-56)
            const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;
            b3 = c2 ? b2 : (c3 ? d : b3);  i3 = c2 ? i2 : (c3 ? k : i3);
            b2 = c1 ? b1 : (c2 ? d : b2);  i2 = c1 ? i1 : (c2 ? k : i2);
            b1 = c1 ? d : b1;              i1 = c1 ? k : i1;
        }
    }
    if (S > 1) {
        sd[w][0][lane] = b1; sd[w][1][lane] = b2; sd[w][2][lane] = b3;
        si[w][0][lane] = i1; si[w][1][lane] = i2; si[w][2][lane] = i3;
        __syncthreads();
        if (w != 0) return;
        // merge: candidates arrive in ascending index order (segment by segment, each list sorted
        // by (d, index)); a candidate replaces on strict '<' only, so equal distances keep the lower
        // index first, as in the sequential scan.  Untouched (inf, 0) entries never insert.
        b1 = b2 = b3 = __builtin_inff(); i1 = i2 = i3 = 0;
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float d = sd[s][q][lane];
                const int k = si[s][q][lane];
                const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;
                b3 = c2 ? b2 : (c3 ? d : b3);  i3 = c2 ? i2 : (c3 ? k : i3);
                b2 = c1 ? b1 : (c2 ? d : b2);  i2 = c1 ? i1 : (c2 ? k : i2);
                b1 = c1 ? d : b1;              i1 = c1 ? k : i1;
            }
    }
    if (valid) {
        float* d = dist2 + ((size_t)bs * n + pt) * 3;
        int32_t* id = idx + ((size_t)bs * n + pt) * 3;
        d[0] = b1; d[1] = b2; d[2] = b3;
        id[0] = i1; id[1] = i2; id[2] = i3;
    }
}

constexpr int TI_CCHUNK = 8;

__global__ __launch_bounds__(256) void three_interpolate_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ idx, const float* __restrict__ weight,
    float* __restrict__ out, int c, int m, int n) {
    const int bs = blockIdx.z;
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const int c0 = blockIdx.y * TI_CCHUNK, c1 = min(c, c0 + TI_CCHUNK);
    const int32_t* id = idx + ((size_t)bs * n + pt) * 3;
    const float* w = weight + ((size_t)bs * n + pt) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float* src = points + ((size_t)bs * c + c0) * m;
    float* dst = out + ((size_t)bs * c + c0) * n + pt;
    for (int ci = c0; ci < c1; ++ci, src += m, dst += n) {
#if PDA_FP_CONTRACT
        *dst = __builtin_fmaf(w2, src[i2], __builtin_fmaf(w1, src[i1], w0 * src[i0]));
#else
        *dst = (w0 * src[i0] + w1 * src[i1]) + w2 * src[i2];
#endif
    }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    const float* __restrict__ grad_out, const int32_t* __restrict__ idx,
    const float* __restrict__ weight, float* __restrict__ grad_points, int c, int n, int m) {
    const int bs = blockIdx.z;
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n) return;
    const int c0 = blockIdx.y * TI_CCHUNK, c1 = min(c, c0 + TI_CCHUNK);
    const int32_t* id = idx + ((size_t)bs * n + pt) * 3;
    const float* w = weight + ((size_t)bs * n + pt) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float* src = grad_out + ((size_t)bs * c + c0) * n + pt;
    float* dst = grad_points + ((size_t)bs * c + c0) * m;
    for (int ci = c0; ci < c1; ++ci, src += n, dst += m) {
        const float g = *src;
        atomicAdd(dst + i0, g * w0);
        atomicAdd(dst + i1, g * w1);
        atomicAdd(dst + i2, g * w2);
    }
}

}  // namespace pda

PDA_API int pda_three_nn(const float* unknown, const float* known, float* dist2, int32_t* idx, int b, int n,
                         int m, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && n >= 0 && m >= 0, "pda_three_nn: negative size");
    if (b == 0 || n == 0) return PDA_OK;
    PDA_REQUIRE(unknown && dist2 && idx && (known || m == 0), "pda_three_nn: null pointer");
    PDA_REQUIRE((int64_t)m * 3 < INT32_MAX && b <= 65535, "pda_three_nn: too large");
    const int tiles = pda::divup(n, 64);
    int S = 1;
    while (S < 8 && (int64_t)tiles * b * S < 4096 && pda::divup(m, S * 2) >= 256) S *= 2;
    const int seglen = pda::divup(pda::divup(m, S), pda::NN_BATCH) * pda::NN_BATCH;
    dim3 grid(tiles, b), block(64 * S);
    hipLaunchKernelGGL(pda::three_nn_kernel, grid, block, 0, (hipStream_t)stream, unknown, known, dist2,
                       idx, n, m, seglen);
    return pda::check_launch("pda_three_nn");
}

PDA_API int pda_three_interpolate(const float* points, const int32_t* idx, const float* weight, float* out,
                                  int b, int c, int m, int n, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, "pda_three_interpolate: negative size");
    if (b == 0 || c == 0 || n == 0) return PDA_OK;
    PDA_REQUIRE(points && idx && weight && out && m > 0, "pda_three_interpolate: null pointer or m == 0");
    PDA_REQUIRE(b <= 65535 && pda::divup(c, pda::TI_CCHUNK) <= 65535, "pda_three_interpolate: too large");
    dim3 grid(pda::divup(n, 256), pda::divup(c, pda::TI_CCHUNK), b), block(256);
    hipLaunchKernelGGL(pda::three_interpolate_kernel, grid, block, 0, (hipStream_t)stream, points, idx,
                       weight, out, c, m, n);
    return pda::check_launch("pda_three_interpolate");
}

PDA_API int pda_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight,
                                       float* grad_points, int b, int c, int n, int m, pda_stream_t stream) {
    PDA_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, "pda_three_interpolate_grad: negative size");
    if (b == 0 || c == 0 || n == 0) return PDA_OK;
    PDA_REQUIRE(grad_out && idx && weight && grad_points && m > 0,
                "pda_three_interpolate_grad: null pointer or m == 0");
    PDA_REQUIRE(b <= 65535 && pda::divup(c, pda::TI_CCHUNK) <= 65535, "pda_three_interpolate_grad: too large");
    dim3 grid(pda::divup(n, 256), pda::divup(c, pda::TI_CCHUNK), b), block(256);
    hipLaunchKernelGGL(pda::three_interpolate_grad_kernel, grid, block, 0, (hipStream_t)stream, grad_out,
                       idx, weight, grad_points, c, n, m);
    return pda::check_launch("pda_three_interpolate_grad");
}
