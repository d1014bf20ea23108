// lz_wave.h -- 64-lane wavefront reductions on DPP (no LDS crossbar): row_shr 1/2/4/8 + row_bcast 15/31,
// result broadcast from lane 63 with v_readlane.  ~6 VALU steps instead of 6 ds_bpermute round trips.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lzw {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float v) {
    return __int_as_float(dpp_i32<CTRL, ROW_MASK>(__float_as_int(old), __float_as_int(v)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    const int lo = dpp_i32<CTRL, ROW_MASK>(__double2loint(old), __double2loint(v));
    const int hi = dpp_i32<CTRL, ROW_MASK>(__double2hiint(old), __double2hiint(v));
    return __hiloint2double(hi, lo);
}

#define LZW_REDUCE(T, DPP, IDENT, OP)                                   \
    v = OP(v, DPP<0x111, 0xf>(IDENT, v)); /* row_shr:1 */               \
    v = OP(v, DPP<0x112, 0xf>(IDENT, v)); /* row_shr:2 */               \
    v = OP(v, DPP<0x114, 0xf>(IDENT, v)); /* row_shr:4 */               \
    v = OP(v, DPP<0x118, 0xf>(IDENT, v)); /* row_shr:8 */               \
    v = OP(v, DPP<0x142, 0xa>(IDENT, v)); /* row_bcast:15 -> rows 1,3 */ \
    v = OP(v, DPP<0x143, 0xc>(IDENT, v)); /* row_bcast:31 -> rows 2,3 */

__device__ __forceinline__ float fmax_sel(float a, float b) { return b > a ? b : a; }
__device__ __forceinline__ double dmax_sel(double a, double b) { return b > a ? b : a; }
__device__ __forceinline__ float fadd(float a, float b) { return a + b; }

// maximum over the wave (NaN inputs are ignored like `x > best` comparisons do)
__device__ __forceinline__ float wave_max(float v) {
    const float ident = -INFINITY;
    LZW_REDUCE(float, dpp_f32, ident, fmax_sel)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_max(double v) {
    const double ident = -INFINITY;
    LZW_REDUCE(double, dpp_f64, ident, dmax_sel)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// sum over the wave (fixed association order, identical on every launch)
__device__ __forceinline__ float wave_sum(float v) {
    const float ident = 0.f;
    LZW_REDUCE(float, dpp_f32, ident, fadd)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef LZW_REDUCE

// The same maximum for inputs that are never NaN, one fused v_max_f32_dpp per step (the select form above costs a DPP
// move + compare + select; `fmaxf` makes the compiler add a canonicalising v_max per operand).  Lanes without a DPP
// source keep their value; two wait states separate a VALU write from a DPP read of the same register.
__device__ __forceinline__ float wave_max_nonan(float v) {
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// inclusive prefix sum over the wave on DPP (row_shr 1/2/4/8 inside the rows of 16, then the row totals with
// row_bcast 15 / 31): 6 VALU steps, no LDS round trips (a __shfl_up ladder is 6 ds_bpermute)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += dpp_i32<0x111, 0xf>(0, v);
    v += dpp_i32<0x112, 0xf>(0, v);
    v += dpp_i32<0x114, 0xf>(0, v);
    v += dpp_i32<0x118, 0xf>(0, v);
    v += dpp_i32<0x142, 0xa>(0, v);
    v += dpp_i32<0x143, 0xc>(0, v);
    return v;
}

// value of `v` in lane `src` (src must be wave-uniform): v_readlane, no LDS
__device__ __forceinline__ int lane_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

}  // namespace lzw
