// Address-space qualified pointer types for the device code: LDS accesses must compile to ds_read/ds_write and
// read-only inputs to scalar/global loads -- never to FLAT instructions (generic pointers kept in structs or passed
// through non-inlined calls defeat the compiler's address-space inference and cost 3-10x per access).
#pragma once
#include "tcv_packed.h"

namespace tcv {
typedef __attribute__((address_space(3))) double lds_d;        // LDS
typedef __attribute__((address_space(3))) int lds_i;
typedef __attribute__((address_space(3))) unsigned lds_u;
typedef __attribute__((address_space(1))) double gbl_d;        // per-workgroup scratch / outputs in HBM
typedef __attribute__((address_space(1))) int gbl_i;
typedef const __attribute__((address_space(4))) double cst_d;  // read-only inputs: window data
typedef const __attribute__((address_space(4))) int cst_i;     // read-only inputs: plan
typedef const __attribute__((address_space(4))) PlanHdr cst_plan;
typedef const __attribute__((address_space(4))) WinHdr cst_win;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4i cst_v4i;
typedef __attribute__((address_space(3))) v4i lds_v4i;

// Sum of a value over the 64 lanes of a wavefront, result valid in lane 0: the tree of `for (o = 32; o > 0; o >>= 1) v += __shfl_down(v, o)`
// -- the same pairings in the same order, hence the same bits (tests/dev/hip/wave_sum_check.hip) -- on the VALU alone: v_permlane32_swap and
// v_permlane16_swap (gfx950) for the two cross-row steps, DPP row_shl for the rest.  __shfl_down is a ds_bpermute per 32 bits: six dependent
// trips through the LDS pipe per reduction.
__device__ __forceinline__ double lane_pair(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
template <int CTRL>
__device__ __forceinline__ double down_dpp(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    return lane_pair(__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double wave_sum_down(double v) {
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        v += lane_pair(__builtin_amdgcn_permlane32_swap(lo, lo, false, false)[1], __builtin_amdgcn_permlane32_swap(hi, hi, false, false)[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        v += lane_pair(__builtin_amdgcn_permlane16_swap(lo, lo, false, false)[1], __builtin_amdgcn_permlane16_swap(hi, hi, false, false)[1]);
    }
    v += down_dpp<0x108>(v); v += down_dpp<0x104>(v); v += down_dpp<0x102>(v); v += down_dpp<0x101>(v);      // row_shl:8, 4, 2, 1
    return v;
}
}  // namespace tcv
#define GEN(p) ((double *)(p))           // explicit address-space cast to generic for the shared factor code (inlined)
#define CGEN(p) ((const double *)(p))
