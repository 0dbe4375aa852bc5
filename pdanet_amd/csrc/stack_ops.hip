// stack_ops.hip -- the pointnet2_stack operator set (include/pda_pointnet2_stack.h): variable-length scenes
// described by *_batch_cnt arrays, point-major (N, C) features.  Never reached by PDA-SSD;
// built so that the reference's pointnet2_stack call sites bind to the same library.
//
// Each workgroup first turns the batch counts into prefix sums in LDS (B is small) instead of the
// reference's per-thread linear scans; queries stage the scene's points through LDS in 256-point tiles
// (coalesced loads, broadcast reads) and a workgroup that straddles a scene boundary walks the scenes
// its lanes belong to.
#include "pda_common.h"

namespace pda {

constexpr int STACK_MAX_B = 1024;

// prefix[k] = cnt[0] + ... + cnt[k-1], k = 0..b (serial: b is the batch size)
__device__ __forceinline__ void stack_prefix(const int* __restrict__ cnt, int b, int* prefix) {
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 0; k < b; ++k) { prefix[k] = acc; acc += cnt[k]; }
        prefix[b] = acc;
    }
}
// scene of global index i: the kernels' scan semantics (ball_query_gpu.cu:27-35): indices past the total
// stay in the last scene
__device__ __forceinline__ int stack_scene(const int* prefix, int b, int i) {
    int bs = 0;
    for (int k = 1; k < b; ++k) bs = (i >= prefix[k]) ? k : bs;
    return bs;
}

template <bool THREE_NN>
__global__ __launch_bounds__(256) void stack_query_kernel(const float* __restrict__ q_xyz, const int* __restrict__ q_cnt,
                                                          const float* __restrict__ xyz, const int* __restrict__ cnt, int b,
                                                          int m, float radius2, int nsample, int* __restrict__ idx,
                                                          float* __restrict__ dist2) {
    __shared__ int qpre[STACK_MAX_B + 1], ppre[STACK_MAX_B + 1];
    __shared__ float tile[256 * 3];
    stack_prefix(q_cnt, b, qpre);
    if (threadIdx.x == 64) {
        int acc = 0;
        for (int k = 0; k < b; ++k) { ppre[k] = acc; acc += cnt[k]; }
        ppre[b] = acc;
    }
    __syncthreads();
    const int pt = blockIdx.x * 256 + threadIdx.x;
    const bool live = pt < m;
    const int first = blockIdx.x * 256, last = min(m, first + 256) - 1;
    const int s_lo = stack_scene(qpre, b, first), s_hi = stack_scene(qpre, b, last);
    const int my_scene = live ? stack_scene(qpre, b, pt) : -1;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) { qx = q_xyz[(size_t)pt * 3 + 0]; qy = q_xyz[(size_t)pt * 3 + 1]; qz = q_xyz[(size_t)pt * 3 + 2]; }
    int found = 0;                                                   // ball query
    double b1 = 1e40, b2 = 1e40, b3 = 1e40;                          // three_nn (best* are double in the reference)
    int i1 = 0, i2 = 0, i3 = 0;
    int* my_idx = idx + (size_t)pt * (THREE_NN ? 3 : nsample);
    for (int s = s_lo; s <= s_hi; ++s) {
        const int start = ppre[s], n = ppre[s + 1] - ppre[s];
        for (int k0 = 0; k0 < n; k0 += 256) {
            const int nk = min(256, n - k0);
            __syncthreads();
            for (int e = threadIdx.x; e < nk * 3; e += 256) tile[e] = xyz[(size_t)(start + k0) * 3 + e];
            __syncthreads();
            if (my_scene != s) continue;
            if (THREE_NN) {
                for (int k = 0; k < nk; ++k) {
                    const float d = sqdist3(qx, qy, qz, tile[k * 3 + 0], tile[k * 3 + 1], tile[k * 3 + 2]);
                    const int g = k0 + k;
                    if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = g; }
                    else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = g; }
                    else if (d < b3) { b3 = d; i3 = g; }
                }
            } else if (found < nsample) {
                for (int k = 0; k < nk; ++k) {
                    const float d2 = sqdist3(qx, qy, qz, tile[k * 3 + 0], tile[k * 3 + 1], tile[k * 3 + 2]);
                    if (d2 < radius2) {
                        if (found == 0)
                            for (int l = 0; l < nsample; ++l) my_idx[l] = k0 + k;
                        my_idx[found] = k0 + k;
                        if (++found >= nsample) break;
                    }
                }
            }
        }
    }
    if (!live) return;
    if (THREE_NN) {
        const int start = ppre[my_scene];
        dist2[(size_t)pt * 3 + 0] = (float)b1; dist2[(size_t)pt * 3 + 1] = (float)b2; dist2[(size_t)pt * 3 + 2] = (float)b3;
        my_idx[0] = i1 + start; my_idx[1] = i2 + start; my_idx[2] = i3 + start;
    } else if (found == 0) {
        my_idx[0] = -1;
    }
}

// out[pt, c, s] = features[start(scene(pt)) + idx[pt, s], c]; GRAD: the scatter-add
template <bool GRAD>
__global__ __launch_bounds__(256) void stack_group_kernel(const float* __restrict__ src, const int* __restrict__ f_cnt,
                                                          const int* __restrict__ idx, const int* __restrict__ i_cnt, int b,
                                                          int m, int c, int nsample, float* __restrict__ dst) {
    __shared__ int ipre[STACK_MAX_B + 1], fpre[STACK_MAX_B + 1];
    stack_prefix(i_cnt, b, ipre);
    if (threadIdx.x == 64) {
        int acc = 0;
        for (int k = 0; k < b; ++k) { fpre[k] = acc; acc += f_cnt[k]; }
        fpre[b] = acc;
    }
    __syncthreads();
    const int64_t total = (int64_t)m * c * nsample;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int s = (int)(e % nsample), ch = (int)((e / nsample) % c), pt = (int)(e / nsample / c);
        const int row = fpre[stack_scene(ipre, b, pt)] + idx[(size_t)pt * nsample + s];
        if (GRAD) atomicAdd(dst + (size_t)row * c + ch, src[e]);
        else dst[e] = src[(size_t)row * c + ch];
    }
}

template <bool GRAD>
__global__ __launch_bounds__(256) void stack_interpolate_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                                const float* __restrict__ weight, int n, int c,
                                                                float* __restrict__ dst) {
    const int64_t total = (int64_t)n * c;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int pt = (int)(e / c), ch = (int)(e % c);
        const int i0 = idx[pt * 3 + 0], i1 = idx[pt * 3 + 1], i2 = idx[pt * 3 + 2];
        const float w0 = weight[pt * 3 + 0], w1 = weight[pt * 3 + 1], w2 = weight[pt * 3 + 2];
        if (GRAD) {
            const float g = src[e];
            atomicAdd(dst + (size_t)i0 * c + ch, g * w0);
            atomicAdd(dst + (size_t)i1 * c + ch, g * w1);
            atomicAdd(dst + (size_t)i2 * c + ch, g * w2);
        } else {
            const float p0 = src[(size_t)i0 * c + ch], p1 = src[(size_t)i1 * c + ch], p2 = src[(size_t)i2 * c + ch];
#if PDA_FP_CONTRACT
            dst[e] = __builtin_fmaf(w2, p2, __builtin_fmaf(w1, p1, w0 * p0));
#else
            dst[e] = (w0 * p0 + w1 * p1) + w2 * p2;
#endif
        }
    }
}

static unsigned grid_for(int64_t total) {
    const int64_t blocks = divup64(total, 256);
    return (unsigned)(blocks < 65536 ? (blocks > 0 ? blocks : 1) : 65536);
}

}  // namespace pda

#define PDA_STACK_B(b, what) PDA_REQUIRE((b) >= 1 && (b) <= pda::STACK_MAX_B, what ": batch size %d outside [1, %d]", (b), pda::STACK_MAX_B)

PDA_API int pda_stack_ball_query(const float* new_xyz, const int32_t* new_xyz_batch_cnt, const float* xyz,
                                 const int32_t* xyz_batch_cnt, int32_t* idx, int b, int m, float radius, int nsample,
                                 pda_stream_t stream) {
    PDA_REQUIRE(m >= 0 && nsample >= 1, "pda_stack_ball_query: m=%d nsample=%d", m, nsample);
    if (m == 0) return PDA_OK;
    PDA_STACK_B(b, "pda_stack_ball_query");
    PDA_REQUIRE(new_xyz && new_xyz_batch_cnt && xyz && xyz_batch_cnt && idx, "pda_stack_ball_query: null pointer");
    hipLaunchKernelGGL(pda::stack_query_kernel<false>, dim3(pda::divup(m, 256)), dim3(256), 0, (hipStream_t)stream, new_xyz,
                       new_xyz_batch_cnt, xyz, xyz_batch_cnt, b, m, radius * radius, nsample, idx, (float*)nullptr);
    return pda::check_launch("pda_stack_ball_query");
}

PDA_API int pda_stack_three_nn(const float* unknown, const int32_t* unknown_batch_cnt, const float* known,
                               const int32_t* known_batch_cnt, float* dist2, int32_t* idx, int b, int n, pda_stream_t stream) {
    PDA_REQUIRE(n >= 0, "pda_stack_three_nn: n=%d", n);
    if (n == 0) return PDA_OK;
    PDA_STACK_B(b, "pda_stack_three_nn");
    PDA_REQUIRE(unknown && unknown_batch_cnt && known && known_batch_cnt && dist2 && idx, "pda_stack_three_nn: null pointer");
    hipLaunchKernelGGL(pda::stack_query_kernel<true>, dim3(pda::divup(n, 256)), dim3(256), 0, (hipStream_t)stream, unknown,
                       unknown_batch_cnt, known, known_batch_cnt, b, n, 0.f, 3, idx, dist2);
    return pda::check_launch("pda_stack_three_nn");
}

PDA_API int pda_stack_group_points(const float* features, const int32_t* features_batch_cnt, const int32_t* idx,
                                   const int32_t* idx_batch_cnt, float* out, int b, int m, int c, int nsample,
                                   pda_stream_t stream) {
    PDA_REQUIRE(m >= 0 && c >= 0 && nsample >= 0, "pda_stack_group_points: bad size");
    if ((int64_t)m * c * nsample == 0) return PDA_OK;
    PDA_STACK_B(b, "pda_stack_group_points");
    PDA_REQUIRE(features && features_batch_cnt && idx && idx_batch_cnt && out, "pda_stack_group_points: null pointer");
    hipLaunchKernelGGL(pda::stack_group_kernel<false>, dim3(pda::grid_for((int64_t)m * c * nsample)), dim3(256), 0,
                       (hipStream_t)stream, features, features_batch_cnt, idx, idx_batch_cnt, b, m, c, nsample, out);
    return pda::check_launch("pda_stack_group_points");
}

PDA_API int pda_stack_group_points_grad(const float* grad_out, const int32_t* idx, const int32_t* idx_batch_cnt,
                                        const int32_t* features_batch_cnt, float* grad_features, int b, int m, int c, int n,
                                        int nsample, pda_stream_t stream) {
    (void)n;
    PDA_REQUIRE(m >= 0 && c >= 0 && nsample >= 0, "pda_stack_group_points_grad: bad size");
    if ((int64_t)m * c * nsample == 0) return PDA_OK;
    PDA_STACK_B(b, "pda_stack_group_points_grad");
    PDA_REQUIRE(grad_out && idx && idx_batch_cnt && features_batch_cnt && grad_features, "pda_stack_group_points_grad: null pointer");
    hipLaunchKernelGGL(pda::stack_group_kernel<true>, dim3(pda::grid_for((int64_t)m * c * nsample)), dim3(256), 0,
                       (hipStream_t)stream, grad_out, features_batch_cnt, idx, idx_batch_cnt, b, m, c, nsample, grad_features);
    return pda::check_launch("pda_stack_group_points_grad");
}

PDA_API int pda_stack_three_interpolate(const float* features, const int32_t* idx, const float* weight, float* out, int n,
                                        int c, pda_stream_t stream) {
    PDA_REQUIRE(n >= 0 && c >= 0, "pda_stack_three_interpolate: bad size");
    if ((int64_t)n * c == 0) return PDA_OK;
    PDA_REQUIRE(features && idx && weight && out, "pda_stack_three_interpolate: null pointer");
    hipLaunchKernelGGL(pda::stack_interpolate_kernel<false>, dim3(pda::grid_for((int64_t)n * c)), dim3(256), 0,
                       (hipStream_t)stream, features, idx, weight, n, c, out);
    return pda::check_launch("pda_stack_three_interpolate");
}

PDA_API int pda_stack_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight,
                                             float* grad_features, int n, int c, pda_stream_t stream) {
    PDA_REQUIRE(n >= 0 && c >= 0, "pda_stack_three_interpolate_grad: bad size");
    if ((int64_t)n * c == 0) return PDA_OK;
    PDA_REQUIRE(grad_out && idx && weight && grad_features, "pda_stack_three_interpolate_grad: null pointer");
    hipLaunchKernelGGL(pda::stack_interpolate_kernel<true>, dim3(pda::grid_for((int64_t)n * c)), dim3(256), 0,
                       (hipStream_t)stream, grad_out, idx, weight, n, c, grad_features);
    return pda::check_launch("pda_stack_three_interpolate_grad");
}
