// Device arithmetic shared by the near-field kernels (nearfield_fast.hip: geometry + general order
// sets; nearfield_simple.hip: the round-lens order set): Cody-Waite sin / cos with scalar-register
// coefficients, reciprocal and reciprocal root by one third-order step, an exact square root
// without range scaling.  Amplitude-type accuracy (a few ulp) unless said otherwise.
#pragma once
#include "nearfield_dev.h"

namespace ml {

// sin and cos of x for |x| < ~1e9: three-constant Cody-Waite reduction with FMA (each step
// rounds once, relative to the already small remainder), fdlibm kernel polynomials.
// Horner step p = z * p + C with the constant in a SCALAR register pair: written as plain C++ the
// compiler materialises every fp64 coefficient with two v_mov_b32 in front of a v_fmac (3 vector
// instructions per step, ~100 extra per sample over the four sincos of a sample); an SGPR
// operand costs two scalar moves instead, which issue beside the vector stream.
__device__ __forceinline__ double horner(double z, double p, double C) {
    asm("v_fma_f64 %0, %1, %0, %2" : "+v"(p) : "v"(z), "s"(C));
    return p;
}

__device__ __forceinline__ void sincos_cw(double x, double &s, double &c) {
    const double k = rint(x * 0.63661977236758138243);          // 2/pi
    double r = fma(-k, 1.57079632679489655800e+00, x);          // pi/2 hi
    r = fma(-k, 6.12323399573676603587e-17, r);                 // pi/2 mid (a third term, 1.5e-33 k, is
                                                                // below 1e-27 for the |k| < 1e6 met here)
    const double z = r * r;
    double ps = horner(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = horner(z, ps, 2.75573137070700676789e-06);
    ps = horner(z, ps, -1.98412698298579493134e-04);
    ps = horner(z, ps, 8.33333333332248946124e-03);
    ps = horner(z, ps, -1.66666666666666324348e-01);
    const double sn = fma(z * r, ps, r);
    double pc = horner(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = horner(z, pc, -2.75573143513906633035e-07);
    pc = horner(z, pc, 2.48015872894767294178e-05);
    pc = horner(z, pc, -1.38888888888741095749e-03);
    pc = horner(z, pc, 4.16666666666666019037e-02);
    const double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)k & 3;
    const double a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// x = r + k pi/2 with |r| <= pi/4 (two-constant Cody-Waite, as above); k fits an int for the
// |x| < ~1e9 this kernel meets
__device__ __forceinline__ void reduce_pio2(double x, double &r, int &k) {
    const double kd = rint(x * 0.63661977236758138243);
    r = fma(-kd, 1.57079632679489655800e+00, x);
    r = fma(-kd, 6.12323399573676603587e-17, r);
    k = (int)kd;
}

// sin and cos of x + kq pi/2 (kq = quadrants already split off a larger angle by reduce_pio2)
__device__ __forceinline__ void sincos_cw_q(double x, int kq, double &s, double &c) {
    const double k = rint(x * 0.63661977236758138243);          // 2/pi
    double r = fma(-k, 1.57079632679489655800e+00, x);          // pi/2 hi
    r = fma(-k, 6.12323399573676603587e-17, r);                 // pi/2 mid (a third term, 1.5e-33 k, is
                                                                // below 1e-27 for the |k| < 1e6 met here)
    const double z = r * r;
    double ps = horner(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = horner(z, ps, 2.75573137070700676789e-06);
    ps = horner(z, ps, -1.98412698298579493134e-04);
    ps = horner(z, ps, 8.33333333332248946124e-03);
    ps = horner(z, ps, -1.66666666666666324348e-01);
    const double sn = fma(z * r, ps, r);
    double pc = horner(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = horner(z, pc, -2.75573143513906633035e-07);
    pc = horner(z, pc, 2.48015872894767294178e-05);
    pc = horner(z, pc, -1.38888888888741095749e-03);
    pc = horner(z, pc, 4.16666666666666019037e-02);
    const double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = ((int)k + kq) & 3;
    const double a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// 1 / sqrt(x) to ~1 ulp for well-scaled x (no denormal / overflow handling): hardware estimate
// (~2^-24) + ONE third-order step, y (1 + e/2 + 3 e^2 / 8) with e = 1 - x y^2 (error ~e^3: five
// operations where two Newton steps take eight).  Amplitude-type quantities only.
__device__ __forceinline__ double rsqrt_fast(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y, e * fma(e, 0.375, 0.5), y);
}

// Accurate reciprocal (~1 ulp): hardware estimate (~2^-24) + one third-order step y (1 + e + e^2),
// e = 1 - x y (three operations where two Newton steps take four).
__device__ __forceinline__ double recip(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, fma(e, e, e), y);
}

// products of two phasors, fused (4 operations): a b and a conj(b)
__device__ __forceinline__ c2 cmulf(c2 a, c2 b) {
    return {fma(a.r, b.r, -(a.i * b.i)), fma(a.r, b.i, a.i * b.r)};
}
__device__ __forceinline__ c2 cmulf_conj(c2 a, c2 b) {
    return {fma(a.r, b.r, a.i * b.i), fma(a.i, b.r, -(a.r * b.i))};
}

// sqrt(x), correctly rounded for well-scaled x (1e-200 < x < 1e200: no denormal / overflow
// handling): hardware reciprocal root, one coupled Goldschmidt step and two residual corrections -
// the sequence the compiler emits for sqrt() minus its range scaling (two ldexp, a class test and
// six selects).  PHASE-CRITICAL use: the propagation distance of nearfield.py:338-341, 453-461,
// whose rounding must be the reference's (np.sqrt is correctly rounded).
__device__ __forceinline__ double sqrt_exact(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
}

}  // namespace ml
