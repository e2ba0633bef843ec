// Shared pieces of the split-operand ("f16x3") kernels (split_x3.hip).
#pragma once
#include "ief_common.h"

#define YBK 32
#ifndef X3_GROUP_M
#define X3_GROUP_M 8
#endif
#define YLD 48          // halves per LDS row: 32 of the K tile + 16 of padding (96 B)

typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct RowCoordY { int b, oy, ox, ok; };

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload(rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

// s x -> (hi, lo) for four consecutive k: v_pk_mul, v_cvt_pk_f16_f32 (round to nearest), then lo = fp16(s x - hi) with the
// subtraction as ONE v_fma_mix_f32 per element (it reads the fp16 half directly: no conversion back) -- 10 VALU per chunk
__device__ __forceinline__ void split4(const f32x4 v, const float s, half4& hi, half4& lo) {
    const f32x4 x = v * s;
    const half2_t h0 = __builtin_convertvector(f32x2{x[0], x[1]}, half2_t);
    const half2_t h1 = __builtin_convertvector(f32x2{x[2], x[3]}, half2_t);
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x[0]), "v"(h0));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x[1]), "v"(h0));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r2) : "v"(x[2]), "v"(h1));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r3) : "v"(x[3]), "v"(h1));
    const half2_t l0 = __builtin_convertvector(f32x2{r0, r1}, half2_t);
    const half2_t l1 = __builtin_convertvector(f32x2{r2, r3}, half2_t);
    hi = half4{h0[0], h0[1], h1[0], h1[1]};
    lo = half4{l0[0], l0[1], l1[0], l1[1]};
}

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of halves, delivered column-major (lane i of the group gets
// column i of the 4 rows); every lane passes the address of row (L >> 2), columns 4 (L & 3) .. of ITS group's block (L = lane & 15).
// EXEC must be all ones.
typedef __fp16 x3_fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) x3_fp16x4_t x3_lds_fp16x4_t;
__device__ __forceinline__ half4 x3_lds_tr_read(const half_t* p) {
    const x3_fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((x3_lds_fp16x4_t*)p);
    return __builtin_bit_cast(half4, v);
}

// one LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses -> 1 KiB of LDS at a wave-uniform base
__device__ __forceinline__ void x3_glds16(const char* g, half_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

enum { X3_LIN = 0, X3_LIN_SLOW = 1, X3_CONV = 2, X3_CONV_UPS = 3, X3_CONV_SLOW = 4 };
