// optim.hip -- the adam_onecycle parameter update as two passes over flat fp32 buffers
// (include/pda_train.h).  The reference steps ~330 tensors one by one from Python
// (fastai_optim.py:138-156: a mul_ per tensor for the decoupled decay, then torch.optim.Adam) after
// clip_grad_norm_ (train_utils.py:56); here the whole model (5.97 M floats = 24 MB) is one HBM-bound
// launch: read p, g, m, v once, write p, m, v once (28 bytes per parameter).
#include "pda_common.h"

namespace pda {

constexpr int NORM_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = g4[i];
        acc += x.x * x.x; acc += x.y * x.y; acc += x.z * x.z; acc += x.w * x.w;
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < (n & 3)) {
        const float x = g[(n4 << 2) + threadIdx.x];
        acc += x * x;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane_id() == 0) red[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void norm_final_kernel(const float* __restrict__ partial, float* __restrict__ norm_out) {
    __shared__ float red[4];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NORM_BLOCKS / 256; ++k) acc += partial[threadIdx.x + 256 * k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane_id() == 0) red[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) norm_out[0] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}

struct AdamArgs {
    float lr, beta1, beta2, eps, decay, step_size, bc2_sqrt, max_norm;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float clip) {
    g *= clip;
    p *= a.decay;
    m = m + (g - m) * (1.f - a.beta1);
    v = v * a.beta2 + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        AdamArgs a, const float* __restrict__ total_norm) {
    float clip = 1.f;
    if (total_norm) clip = fminf(1.f, a.max_norm / (total_norm[0] + 1e-6f));
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 P = reinterpret_cast<float4*>(p)[i], M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        adam_one(P.x, G.x, M.x, V.x, a, clip);
        adam_one(P.y, G.y, M.y, V.y, a, clip);
        adam_one(P.z, G.z, M.z, V.z, a, clip);
        adam_one(P.w, G.w, M.w, V.w, a, clip);
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        adam_one(p[i], g[i], m[i], v[i], a, clip);
    }
}

}  // namespace pda

PDA_API int pda_grad_norm(const float* g, int64_t n, float* norm_out, float* scratch1024, pda_stream_t stream) {
    PDA_REQUIRE(n >= 0, "pda_grad_norm: n = %lld", (long long)n);
    PDA_REQUIRE(norm_out && scratch1024 && (g || n == 0), "pda_grad_norm: null pointer");
    PDA_REQUIRE(((uintptr_t)g & 15) == 0, "pda_grad_norm: buffer must be 16-byte aligned");
    hipLaunchKernelGGL(pda::sumsq_partial_kernel, dim3(pda::NORM_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, n, scratch1024);
    hipLaunchKernelGGL(pda::norm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch1024, norm_out);
    return pda::check_launch("pda_grad_norm");
}

PDA_API int pda_adam_onecycle_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                   float beta2, float eps, float wd, int step, const float* total_norm,
                                   float max_norm, pda_stream_t stream) {
    PDA_REQUIRE(n >= 0 && step >= 1, "pda_adam_onecycle_step: n = %lld, step = %d", (long long)n, step);
    if (n == 0) return PDA_OK;
    PDA_REQUIRE(p && g && m && v, "pda_adam_onecycle_step: null pointer");
    PDA_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                "pda_adam_onecycle_step: buffers must be 16-byte aligned");
    PDA_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "pda_adam_onecycle_step: betas out of range");
    // scalars in double like torch's Python-side scalars (adam.py _single_tensor_adam)
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    pda::AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.decay = (float)(1.0 - (double)wd * (double)lr);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.max_norm = max_norm;
    const int64_t n4 = (n + 3) / 4;
    const int64_t blocks = n4 / 256 + 1;
    hipLaunchKernelGGL(pda::adam_step_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, p, g, m, v, n, a, total_norm);
    return pda::check_launch("pda_adam_onecycle_step");
}
