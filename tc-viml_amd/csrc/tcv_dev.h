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
}  // namespace tcv
#define GEN(p) ((double *)(p))           // explicit address-space cast to generic for the shared factor code (inlined)
#define CGEN(p) ((const double *)(p))
