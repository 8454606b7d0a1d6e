// wide_common.h -- LDS-direct load and hand-scheduled LDS read helpers shared by the weight-streaming
// encoder kernels (wide_kernels.hip: one launch per layer; wide_fused_kernels.hip: the whole encoder in one).
#pragma once

#include "encoder_core.h"

namespace qbw {

using qb::f16x8;
using qb::f32x4;

__device__ __forceinline__ f16x8 as_frag(const uint4& u) { return __builtin_bit_cast(f16x8, u); }

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    // LDS-direct load: lane l's 16 bytes at src land at lds_dst + 16 l (lds_dst is wave-uniform)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// LDS reads of the staged images go through inline asm: the compiler orders an LDS load it can see
// behind EVERY LDS-direct load in flight (s_waitcnt vmcnt(0): it cannot tell the ring slots apart),
// which would drain the activation stream at every k-step.  An asm read is invisible to that
// bookkeeping, so its completion is waited for explicitly (lds_wait ties the s_waitcnt to the
// registers, which keeps consumers behind it).
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // one VGPR quad (asm "v" operand)
template <int OFF>
__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read_b128 immediate offset is 16 bits");
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
__device__ __forceinline__ void lds_wait(u32x4& a) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a));
}
__device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}


}  // namespace qbw
