// pda_common.h -- shared host/device helpers for libpda_pointnet2.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/pda_pointnet2.h"
#include "../../include/pda_train.h"
#include "../../include/pda_pointnet2_stack.h"

#ifndef PDA_FP_CONTRACT
#define PDA_FP_CONTRACT 1  // 1: fma(dz,dz,fma(dy,dy,dx*dx)) (nvcc -fmad=true); 0: uncontracted
#endif

#define PDA_API extern "C" __attribute__((visibility("default")))
#define PDA_WAVE 64

namespace pda {

// ---- host-side status plumbing ---------------------------------------------------------
void set_error(const char* fmt, ...);


#define PDA_REQUIRE(cond, ...)                                      \
    do {                                                            \
        if (!(cond)) {                                              \
            ::pda::set_error(__VA_ARGS__);                          \
            return PDA_ERR_INVALID_ARGUMENT;                        \
        }                                                           \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(err));
        return PDA_ERR_LAUNCH;
    }
    return PDA_OK;
}

// Once-per-DEVICE host-side initialisation (a function attribute, an occupancy query): function-local statics are
// evaluated for whichever device is current first, but a process may drive several devices.  `slot` is a function-local
// `static pda::PerDevice<T>`; get() runs `init` the first time the CURRENT device asks (races re-run it: idempotent).
template <typename T> struct PerDevice {
    static constexpr int MAX_DEV = 64;
    T value[MAX_DEV];
    std::atomic<bool> done[MAX_DEV];
    PerDevice() { for (int i = 0; i < MAX_DEV; ++i) done[i].store(false); }
    template <typename F> T get(F init) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return init();
        if (!done[dev].load(std::memory_order_acquire)) {
            value[dev] = init();
            done[dev].store(true, std::memory_order_release);
        }
        return value[dev];
    }
};

inline int divup(int a, int b) { return (a + b - 1) / b; }
inline int64_t divup64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers --------------------------------------------------------------------
// Squared distance (a - b), the one expression every query kernel shares
// (ball_query_gpu.cu:33, interpolate_gpu.cu:43, sampling_gpu.cu:133).  The file is built
// with -ffp-contract=off, so the only contraction is the one written here.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by,
                                         float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
#if PDA_FP_CONTRACT
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
#else
    return (dx * dx + dy * dy) + dz * dz;
#endif
}

// wave-uniform wave index inside the workgroup, provably uniform to the compiler
// Tensor element type at the HBM boundary: float, or bf16 (raw uint16_t) in the dense-bf16 mode, where qkv / dout come
// straight out of bf16 GEMMs and out / dqkv go straight into them.  All arithmetic is fp32 either way.
typedef uint16_t bf16_t;
__device__ __forceinline__ float bf16_to_f32(bf16_t u) { return __uint_as_float((uint32_t)u << 16); }
// f32 -> bf16 through the hardware conversion (v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN).  The usual
// integer recipe (u + 0x7fff + ((u >> 16) & 1)) >> 16 turns some NaNs into 0 or infinity, and a guard "f != f" does not
// survive -fno-honor-nans (MI355X_MICROARCH.md, correctness boundaries).
typedef __bf16 pda_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pda_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
    const pda_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pda_bf16x2));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(f32x2_to_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void store4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, const float4& v) {
    uint2 u;
    u.x = f32x2_to_bf16x2(v.x, v.y);
    u.y = f32x2_to_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const bf16_t* p) { return bf16_to_f32(*p); }

__device__ __forceinline__ int wave_id() {
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Read-only, wave-uniform streams (the point cloud every lane tests against) are read
// through the CONSTANT address space: with a wave-uniform address the backend then emits
// scalar loads (s_load_dwordxN into SGPRs) instead of 64-lane vector loads.  Only valid for
// buffers no kernel in flight writes (xyz / known / new_xyz inputs).
typedef const float __attribute__((address_space(4))) * cfloat_ptr;
__device__ __forceinline__ cfloat_ptr as_constant(const float* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (cfloat_ptr)p;
#pragma clang diagnostic pop
}

// Forces a pointer the program knows to be wave-uniform into SGPRs (two v_readfirstlane);
// divergence analysis loses uniformity across PHIs that join divergent control flow.
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T*)(((uint64_t)hi << 32) | lo);
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).  Unlike
// __syncthreads() it does NOT wait for outstanding global stores/loads (vmcnt), so a result store
// issued in a latency-critical loop (FPS writes idx[j] every round) stays in flight across it.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Wave-wide reductions of idempotent ops with the DPP modifier fused into the ALU op
// (gfx9 DPP: row_shr:n inside rows of 16 lanes, then row_bcast:15 / row_bcast:31 across
// rows).  A lane whose DPP source is out of range keeps its own value.  `s_nop 1` covers
// the 2-wait-state VALU-write -> DPP-read hazard, which hipcc does not pad inside asm.
// Result: valid in lane 63, returned wave-uniform through readlane.
#define PDA_DPP_REDUCE64(OP)                                                   \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"  \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"  \
    "s_nop 1"
// Same over the first 16 lanes only (row 0); result valid in lane 15.
#define PDA_DPP_REDUCE16(OP)                                                   \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1"

// All-reduce inside each row of 16 lanes (row rotations): every lane gets its row's result.
#define PDA_DPP_ROW_ALLREDUCE(OP)                                              \
    "s_nop 1\n\t" OP " %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1"
__device__ __forceinline__ float row_allmax_f32(float v) {
    asm volatile(PDA_DPP_ROW_ALLREDUCE("v_max_f32_dpp") : "+v"(v));
    return v;
}
__device__ __forceinline__ uint32_t row_allmin_u32(uint32_t v) {
    asm volatile(PDA_DPP_ROW_ALLREDUCE("v_min_u32_dpp") : "+v"(v));
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
    asm volatile(PDA_DPP_REDUCE64("v_max_f32_dpp") : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// v[lane] = -1.0f for one wave-uniform lane (v_writelane_b32 with an inline constant: a second SGPR operand would break
// the constant-bus limit; hipcc 7.2 has no builtin for the instruction).
__device__ __forceinline__ float writelane_minus_one(int lane_uniform, float v) {
    asm volatile("v_writelane_b32 %0, -1.0, %1" : "+v"(v) : "s"(lane_uniform));
    return v;
}
// Two independent maxima in one pass: the second chain fills the wait states of the first (2 x 6 DPP operations and six
// s_nop 0 instead of 2 x (6 + six s_nop 1)).
#define PDA_DPP_STAGE_X2(OP, CTRL)                                             \
    OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
__device__ __forceinline__ void wave_max2_f32(float& a, float& b) {
    asm volatile("s_nop 1\n\t"
                 PDA_DPP_STAGE_X2("v_max_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")
                 PDA_DPP_STAGE_X2("v_max_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
                 PDA_DPP_STAGE_X2("v_max_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
                 PDA_DPP_STAGE_X2("v_max_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
                 PDA_DPP_STAGE_X2("v_max_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
                 PDA_DPP_STAGE_X2("v_max_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(a), "+v"(b));
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
    b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    asm volatile(PDA_DPP_REDUCE64("v_min_u32_dpp") : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float row0_max_f32(float v) {
    asm volatile(PDA_DPP_REDUCE16("v_max_f32_dpp") : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 15));
}
__device__ __forceinline__ uint32_t row0_min_u32(uint32_t v) {
    asm volatile(PDA_DPP_REDUCE16("v_min_u32_dpp") : "+v"(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}

}  // namespace pda
