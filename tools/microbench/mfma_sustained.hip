// Sustained v_mfma_f32_32x32x16_bf16 rate of the whole chip: nothing but matrix instructions on register operands, so
// the number is what the power-managed clock leaves of the 2.5 PFLOP/s nominal peak (MI355X_MICROARCH.md) for a kernel
// whose matrix pipes never wait.  Operands: zeros, or random normal values (switching activity costs clock).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_sustained.hip -o build/mfma_sustained && build/mfma_sustained
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ACC>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(i * 64 + lane) & 1023]);
        b[i] = __builtin_bit_cast(bf16x8, src[(128 + i * 64 + lane) & 1023]);
    }
    f32x16 acc[ACC];
    for (int k = 0; k < ACC; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k & 1], b[(k >> 1) & 1], acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < ACC; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;     // never true: keeps the loop
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    std::vector<unsigned short> h(1024 * 8);
    uint4* src; float* out;
    CK(hipMalloc(&src, 1024 * 16)); CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%s, %d CUs, clockRate %.2f GHz\n", p.name, cus, p.clockRate / 1e6);
    for (int data = 0; data < 2; ++data) {
        srand(1);
        for (auto& v : h) {
            float f = 0.f;
            if (data) { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX; f = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
            unsigned int bits; memcpy(&bits, &f, 4); v = (unsigned short)(bits >> 16);
        }
        CK(hipMemcpy(src, h.data(), 1024 * 16, hipMemcpyHostToDevice));
        for (int wpc = 4; wpc <= 8; wpc += 4) {               // waves per CU: 1 or 2 per SIMD
            const int blocks = cus * wpc / 4, iters = 20000;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double flops = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
                if (rep == 2)
                    printf("%-14s %d wave(s)/SIMD: %.2f ms  %.0f TFLOP/s  (%.2f of 2500 nominal; implied matrix clock %.2f GHz)\n",
                           data ? "random normal" : "zeros", wpc / 4, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0,
                           flops / (ms * 1e-3) / ((double)cus * 4 * 1024.0) / 1e9);
            }
        }
    }
    return 0;
}
