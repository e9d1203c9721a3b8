// afg_pk.h -- packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32) for the transform kernels.
//
// One instruction works on a 64-bit register pair.  op_sel / op_sel_hi pick which half of each source
// feeds the low / high result, neg_lo / neg_hi flip the sign of a source half.  Every product and sum is
// rounded exactly as the scalar expression it replaces (a*(-b) == -(a*b), a + (-b) == a - b, -(a-b) ==
// b-a), so results stay bit-identical to the reference's expression trees.  The compiler folds whole-
// pair swizzles and negations of plain vector code by itself; mixed forms are spelled out here.
//
// Naming: pk_<op>_<lo>_<hi>, each half written as <src0 half><src1 half> with l = .x, h = .y and an n in
// front of a negated factor: pk_mul_ll_hl(a, b) = (a.x*b.x, a.y*b.x), pk_add_lh_hnl = (a.x+b.y, a.y-b.x).
#pragma once

typedef float f2 __attribute__((ext_vector_type(2)));

#define AFG_PK(name, op, mods)                                                              \
    __device__ __forceinline__ f2 name(f2 a, f2 b)                                          \
    {                                                                                       \
        f2 r;                                                                               \
        asm(op " %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b));                             \
        return r;                                                                           \
    }
AFG_PK(pk_mul_ll_hl, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[1,0]")                        // ( a.x*b.x ,  a.y*b.x)
AFG_PK(pk_mul_xneg, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[1,0]")            // (-a.y*b.y ,  a.x*b.y)
AFG_PK(pk_mul_lh_ll, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[0,0]")                        // ( a.x*b.y ,  a.x*b.x)
AFG_PK(pk_mul_ll_lnh, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]")          // ( a.x*b.x , -a.x*b.y)
AFG_PK(pk_mul_nhh_nhl, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]")   // (-a.y*b.y , -a.y*b.x)
AFG_PK(pk_mul_nhl_hh, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0]")          // (-a.y*b.x ,  a.y*b.y)
AFG_PK(pk_mul_hl_ll, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[0,0]")                        // ( a.y*b.x ,  a.x*b.x)
AFG_PK(pk_mul_nlh_hh, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[1,0]")          // (-a.x*b.y ,  a.y*b.y)
AFG_PK(pk_mul_lh_hh, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,1]")                        // ( a.x*b.y ,  a.y*b.y)
AFG_PK(pk_mul_hl_nll, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]")          // ( a.y*b.x , -a.x*b.x)
AFG_PK(pk_mul_lh_nll, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[1,0]")          // ( a.x*b.y , -a.x*b.x)
AFG_PK(pk_mul_nhl_nhh, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]")   // (-a.y*b.x , -a.y*b.y)
AFG_PK(pk_add_swap, "v_pk_add_f32", "op_sel:[1,1] op_sel_hi:[0,0]")                         // ( a.y+b.y ,  a.x+b.x)
AFG_PK(pk_add_lh_hnl, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")          // ( a.x+b.y ,  a.y-b.x)
AFG_PK(pk_add_hnl_lh, "v_pk_add_f32", "op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1]")          // ( a.y-b.x ,  a.x+b.y)
AFG_PK(pk_add_hnh_nll, "v_pk_add_f32", "op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]")   // ( a.y-b.y , -a.x+b.x)
AFG_PK(pk_add_lnh_hl, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")          // ( a.x-b.y ,  a.y+b.x)
AFG_PK(pk_add_lnl_hh, "v_pk_add_f32", "op_sel:[0,0] op_sel_hi:[1,1] neg_lo:[0,1]")          // ( a.x-b.x ,  a.y+b.y)
AFG_PK(pk_add_ll_hnh, "v_pk_add_f32", "op_sel:[0,0] op_sel_hi:[1,1] neg_hi:[0,1]")          // ( a.x+b.x ,  a.y-b.y)
AFG_PK(pk_add_lnl_nhh, "v_pk_add_f32", "op_sel:[0,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]")   // ( a.x-b.x , -a.y+b.y)
AFG_PK(pk_mul_ll_lh, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[0,1]")                        // ( a.x*b.x ,  a.x*b.y)
AFG_PK(pk_mul_hl_hh, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[1,1]")                        // ( a.y*b.x ,  a.y*b.y)
AFG_PK(pk_mul_lnh_ll, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]")          // (-a.x*b.y ,  a.x*b.x)
AFG_PK(pk_mul_hnh_hl, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]")          // (-a.y*b.y ,  a.y*b.x)
#undef AFG_PK

// three-operand forms (fused: tolerance-mode kernels only)
#define AFG_PK3(name, op, mods)                                                             \
    __device__ __forceinline__ f2 name(f2 a, f2 b, f2 c)                                    \
    {                                                                                       \
        f2 r;                                                                               \
        asm(op " %0, %1, %2, %3 " mods : "=v"(r) : "v"(a), "v"(b), "v"(c));                 \
        return r;                                                                           \
    }
AFG_PK3(pk_fma_nhh_hl, "v_pk_fma_f32", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]")   // (-a.y*b.y + c.x ,  a.y*b.x + c.y)
#undef AFG_PK3

// complex product in two packed instructions (fused multiply-adds: not the reference's rounding)
__device__ __forceinline__ f2 pk_cmul_fused(f2 a, f2 w) { return pk_fma_nhh_hl(a, w, pk_mul_ll_lh(a, w)); }
