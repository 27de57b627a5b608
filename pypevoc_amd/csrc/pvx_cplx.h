// pvx_cplx.h -- complex float32 arithmetic on register pairs for the in-register FFTs.
//
// A complex value lives in an even-aligned VGPR pair (re, im) so that one packed instruction
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two float32 lanes per issue slot on gfx950) does a
// complex add or half a complex multiply.  Plain +, -, * on the vector type and the broadcast
// swizzles (.xx, .yy) compile to single packed instructions.  What the compiler does NOT fold is a
// swap or a one-sided negation into the packed instruction's op_sel / neg modifiers (it emits
// v_xor + v_mov to build the swizzled operand first), and those are exactly the FFT's multiply-by-(-i)
// and complex-conjugate patterns -- so these few are spelled as one instruction each.
// Hazard rule: the compiler's hazard recogniser does not see inside inline asm, so the result of one of
// these primitives must not be the direct source of a DPP move (VALU write -> DPP read needs two wait
// states); in the FFTs every exchanged value is the result of a compiler-emitted instruction.
// VOP3P modifiers: op_sel[i] / op_sel_hi[i] pick the half (0 = .x, 1 = .y) of source i that feeds the
// low / high result lane; neg_lo[i] / neg_hi[i] negate source i in the low / high lane.
#pragma once

namespace pvxc {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2f mk(float re, float im) { return (v2f){re, im}; }
__device__ __forceinline__ v2f splat(float v) { return (v2f){v, v}; }

// b + (d.y, -d.x) = b - i d
__device__ __forceinline__ v2f add_mni(v2f b, v2f d) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(b), "v"(d));
    return r;
}
// b + (-d.y, d.x) = b + i d
__device__ __forceinline__ v2f add_pi(v2f b, v2f d) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(b), "v"(d));
    return r;
}
// a + conj(b) = (a.x + b.x, a.y - b.y)
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a - conj(b) = (a.x - b.x, a.y + b.y)
__device__ __forceinline__ v2f sub_conj(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (z.y, -z.x) = -i z, exact including the sign of zeros (a multiply by (1, -1), not "0 - x")
__device__ __forceinline__ v2f mni(v2f z) {
    v2f r;
    const v2f c = (v2f){1.f, -1.f};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(z), "s"(c));
    return r;
}
// conj(z)
__device__ __forceinline__ v2f conj(v2f z) { return z * (v2f){1.f, -1.f}; }

// z * w with the rounding of pvxf::cmul: re = fma(z.x, w.x, -(z.y w.y)), im = fma(z.x, w.y, z.y w.x).
// First instruction: t = (z.y * -w.y, z.y * w.x); second: fma(z.xx, w, t).  cmul takes w in VGPRs
// (lane-dependent twiddles), cmul_k in SGPRs (wave-uniform constants).
__device__ __forceinline__ v2f cmul(v2f z, v2f w) {
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(z), "v"(w));
    return __builtin_elementwise_fma(z.xx, w, t);
}
__device__ __forceinline__ v2f cmul_k(v2f z, v2f w) {
    v2f t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(z), "s"(w));
    return __builtin_elementwise_fma(z.xx, w, t);
}

// z * conj(w): re = fma(z.x, w.x, z.y w.y), im = fma(z.x, -w.y, z.y w.x)  (k_fused_team.hip: the twiddles of the
// mirrored half of a split transform are the conjugates of the ones the lane already holds)
__device__ __forceinline__ v2f cmul_conj(v2f z, v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(z), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(z), "v"(w), "v"(t));
    return r;
}

// (c.x * z.y, c.y * z.x): with c = (h, -h) this is -i h z
__device__ __forceinline__ v2f mul_swap(v2f z, v2f c) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(z), "s"(c));
    return r;
}
// conj(h * s - p) = (h s.x - p.x, -(h s.y) + p.y), h = (h, h) uniform
__device__ __forceinline__ v2f fms_conj(v2f h, v2f s, v2f p) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "s"(h), "v"(s), "v"(p));
    return r;
}

// a * s + b, s scalar
__device__ __forceinline__ v2f fma_s(float s, v2f a, v2f b) { return __builtin_elementwise_fma(splat(s), a, b); }

}  // namespace pvxc
