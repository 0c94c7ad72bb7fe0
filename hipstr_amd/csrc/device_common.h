// device_common.h — wave primitives and the bit-exact float log-sum-exp helpers shared by the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace {

constexpr double IMP = HS_IMPOSSIBLE;
constexpr double T_I2I = -1.0, T_I2M = -0.4586751453870818910216436;   // AlignmentModel.h:7-10
constexpr double T_D2D = -1.0, T_D2M = -0.4586751453870818910216436;

// ------------------------------------------------------------------ wave primitives
__device__ __forceinline__ int shr1(int old, int src){
  return __builtin_amdgcn_update_dpp(old, src, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ double shr1(double old, double src){
  const int lo = shr1(__double2loint(old), __double2loint(src));
  const int hi = shr1(__double2hiint(old), __double2hiint(src));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rdlane(int v, int l){ return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double rdlane(double v, int l){
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ uint64_t rdlane(uint64_t v, int l){
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
__device__ __forceinline__ int uni(int v){ return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni(int64_t v){
  return ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ double uni(double v){
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ void wave_lds_sync(){
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Integer minimum / maximum over the wavefront with DPP row shifts and row broadcasts (six VALU operations and a v_readlane) instead of
// six ds_bpermute round trips: these sit on the critical path of the STR kernels' read-end sums.  A lane without a source keeps its own
// value (old = v), which min and max do not mind.
#define HS_DPP_RED(OP) \
  v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false)); /* row_shr:1 */ \
  v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false)); /* row_shr:2 */ \
  v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false)); /* row_shr:4 */ \
  v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false)); /* row_shr:8: lane 15 of every row holds the row */ \
  v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false)); /* row_bcast:15 into rows 1 and 3 */ \
  v = OP(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false)); /* row_bcast:31 into rows 2 and 3: lane 63 holds the wavefront */ \
  return __builtin_amdgcn_readlane(v, 63);
__device__ __forceinline__ int wave_min_i(int v){ HS_DPP_RED(min) }
__device__ __forceinline__ int wave_max_i(int v){ HS_DPP_RED(max) }
#undef HS_DPP_RED
// The same for doubles: two DPP moves and one FP64 operation per step.  The sum is used for log-sum-exp totals only: every term is a
// float (2^-10 <= term <= 1, LOG_THRESH = ln 0.001) converted to double, so any order of additions is exact and gives the same bits.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_d(double v, double old){
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_d(double v){
  v = fmax(v, dpp_d<0x111, 0xf>(v, v)); v = fmax(v, dpp_d<0x112, 0xf>(v, v)); v = fmax(v, dpp_d<0x114, 0xf>(v, v)); v = fmax(v, dpp_d<0x118, 0xf>(v, v));
  v = fmax(v, dpp_d<0x142, 0xa>(v, v)); v = fmax(v, dpp_d<0x143, 0xc>(v, v));
  return rdlane(v, 63);
}
__device__ __forceinline__ double wave_sum_d(double v){
  v += dpp_d<0x111, 0xf>(v, 0.0); v += dpp_d<0x112, 0xf>(v, 0.0); v += dpp_d<0x114, 0xf>(v, 0.0); v += dpp_d<0x118, 0xf>(v, 0.0);   // lane 15 of a row: the row
  v += dpp_d<0x142, 0xa>(v, 0.0);                    // rows 1 and 3 take lane 15 of the row before
  v += dpp_d<0x143, 0xc>(v, 0.0);                    // rows 2 and 3 take lane 31: lane 63 holds the wavefront
  return rdlane(v, 63);
}

// ------------------------------------------------------------------ float approximations (bit-exact)
__device__ __forceinline__ float f_fasterexp(float p){           // fastonebigheader.h:206-218
  const float y = __fmul_rn(1.442695040f, p);
  const float c = (y < -126.0f) ? -126.0f : y;
  return __uint_as_float((uint32_t)__fmul_rn(8388608.0f, __fadd_rn(c, 126.94269504f)));
}
__device__ __forceinline__ float f_fasterlog(float x){           // fastonebigheader.h:348-358
  float y = (float)__float_as_uint(x);
  y = __fmul_rn(y, 8.2629582881927490e-8f);
  return __fsub_rn(y, 87.989971088f);
}

// streaming form of fast_log_sum_exp(vector) (mathops.cpp:97-106): pass 0 finds the max,
// pass 1 accumulates.  The float terms are summed in double, which is exact for any order.
struct Lse {
  double mx, tot;
  __device__ __forceinline__ void start(int pass, double first){ if (pass == 0) mx = first; else tot = 0.0; }
  __device__ __forceinline__ void push(int pass, double v, double thr){
    if (pass == 0) mx = fmax(mx, v);
    else { const double d = v - mx; if (d > thr) tot += (double)f_fasterexp((float)d); }
  }
  __device__ __forceinline__ double finish() const { return mx + (double)f_fasterlog((float)tot); }
};

__device__ __forceinline__ double emit(uint8_t r, uint8_t c, double2 q){ return r == c ? q.x : q.y; }

}  // namespace
