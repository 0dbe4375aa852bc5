// split_bf16.h -- what the split-bf16 matrix kernels share (csrc/gemm_split.hip, csrc/wgrad.hip): the exact three-term
// bf16 split of an f32 value and the LDS-DMA instruction with hand-counted waits.
#pragma once
#include "pda_common.h"

namespace pda {

typedef float gs_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));

// One 16-byte LDS-DMA per lane: global (per-lane address) -> LDS (wave-uniform byte address in M0 + lane * 16), no VGPR
// in between.  Written as asm so that hipcc does not know about the pending LDS write: with the builtin it puts
// s_waitcnt vmcnt(0) in front of every ds_read of the ring, which serialises the DMA of tile t + 2 with the MFMAs of
// tile t.  The waits are the counted ones in the kernel.
__device__ __forceinline__ void glds16(const uint4* g, uint32_t lds_wave_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_wave_base) : "memory");
}

__device__ __forceinline__ void split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = f32x2_to_bf16x2(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = f32x2_to_bf16x2(r0, r1);
    l = f32x2_to_bf16x2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}

}  // namespace pda
