// c8_math.hpp -- scalar and small-tensor types shared by every kernel.
//
// Dual: the flat fixed-size forward-AD type that replaces Sacado::Fad::SLFad
// (reference defines.hpp:23-26).  It carries ONE tangent direction; a lane
// group of a wavefront carries one direction per lane (lane = derivative
// slot), so an N-derivative dual number is spread over N lanes and every
// operation below is one or two FP64 VALU instructions per lane.
//
// Tens3: 3x3 tensor with compile-time indices only (no run-time indexed
// register arrays), the MiniTensor subset of SURVEY.md section 8a row a15.
#pragma once

#include <math.h>

#include <type_traits>
#include <utility>

#if defined(__HIPCC__)
#define C8_HD __host__ __device__ __forceinline__
#else
#define C8_HD inline
#endif
// every small fixed-trip loop must be fully unrolled on the GPU so that register
// arrays are only ever indexed by compile-time constants (no scratch memory)
// stage traffic is written once and read once much later: keep it out of the caches (nontemporal on the GPU)
#if defined(__HIP_DEVICE_COMPILE__)
#define C8_STREAM_STORE(p, v) __builtin_nontemporal_store((v), (p))
#define C8_STREAM_LOAD(p) __builtin_nontemporal_load(p)
// two adjacent doubles (16-byte aligned) with one 16-byte load
#define C8_STREAM_LOAD2(p, a, b)                                                        \
  do {                                                                                  \
    typedef double c8_d2 __attribute__((ext_vector_type(2)));                           \
    c8_d2 const w__ = __builtin_nontemporal_load(reinterpret_cast<c8_d2 const*>(p));    \
    (a) = w__.x;                                                                        \
    (b) = w__.y;                                                                        \
  } while (0)
#else
#define C8_STREAM_STORE(p, v) (*(p) = (v))
#define C8_STREAM_LOAD(p) (*(p))
#define C8_STREAM_LOAD2(p, a, b) do { (a) = (p)[0]; (b) = (p)[1]; } while (0)
#endif
// the instruction scheduler does not move anything across this point (device builds; nothing on the host): keeps the
// operands of one step of an unrolled loop from being fetched steps ahead, which costs registers
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define C8_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// ... and the value x is complete at this point of the instruction stream (an empty asm statement that takes and returns
// it): together with the fence this keeps the arithmetic of one step of an unrolled loop between its two fences
#define C8_PIN(x) asm volatile("" : "+v"(x))
#else
#define C8_SCHED_FENCE() ((void)0)
#define C8_PIN(x) ((void)0)
#endif
#if defined(__clang__)
#define C8_UNROLL _Pragma("unroll")
#define C8_NOUNROLL _Pragma("clang loop unroll(disable)")
#else
#define C8_UNROLL
#define C8_NOUNROLL
#endif

namespace c8 {

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>).  Used where a run-time
// value is compared against the loop index, so that no optimisation pass can fold the
// "search" loop back into a run-time index into a register array.
template <class F, int... I> C8_HD void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> C8_HD void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct Dual {
  double v, d;
  C8_HD Dual() {}
  C8_HD Dual(double x) : v(x), d(0.) {}
  C8_HD Dual(double x, double dx) : v(x), d(dx) {}
};

C8_HD double val(double x) { return x; }
C8_HD double val(Dual const& x) { return x.v; }
C8_HD double der(double) { return 0.; }
C8_HD double der(Dual const& x) { return x.d; }

// 1 / x on the device: v_rcp_f64 and two Newton steps -- accurate to the last place but not correctly rounded, and half the
// dependent instructions of the IEEE division sequence (div_scale x 2, rcp, 4-5 fma, div_fmas, div_fixup); the reciprocals
// sit on the critical path of every evaluation of a model (11.00 against 11.12 ms per assembly, gpurun_out/tune_rcp.log).
// 0 and infinities give NaN instead of inf / 0: the callers divide by zero only where IEEE gives NaN as well (0 / 0 on
// stress-free points, whose branch does not use the quotient).  C8_TUNE_IEEE_DIV: tuning build with the IEEE sequence.
C8_HD double c8_rcp(double x) {
#if !defined(C8_TUNE_IEEE_DIV) && defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.), r, r);
  r = fma(fma(-x, r, 1.), r, r);
  return r;
#else
  return 1. / x;
#endif
}
C8_HD Dual operator-(Dual const& a) { return Dual(-a.v, -a.d); }
C8_HD Dual operator+(Dual const& a, Dual const& b) { return Dual(a.v + b.v, a.d + b.d); }
C8_HD Dual operator-(Dual const& a, Dual const& b) { return Dual(a.v - b.v, a.d - b.d); }
C8_HD Dual operator*(Dual const& a, Dual const& b) { return Dual(a.v * b.v, a.v * b.d + a.d * b.v); }
C8_HD Dual operator/(Dual const& a, Dual const& b) {
  double const r = c8_rcp(b.v);
  double const q = a.v * r;
  return Dual(q, (a.d - q * b.d) * r);
}
C8_HD Dual operator+(Dual const& a, double b) { return Dual(a.v + b, a.d); }
C8_HD Dual operator+(double a, Dual const& b) { return Dual(a + b.v, b.d); }
C8_HD Dual operator-(Dual const& a, double b) { return Dual(a.v - b, a.d); }
C8_HD Dual operator-(double a, Dual const& b) { return Dual(a - b.v, -b.d); }
C8_HD Dual operator*(Dual const& a, double b) { return Dual(a.v * b, a.d * b); }
C8_HD Dual operator*(double a, Dual const& b) { return Dual(a * b.v, a * b.d); }
C8_HD Dual operator/(Dual const& a, double b) {
  double const r = c8_rcp(b);
  return Dual(a.v * r, a.d * r);
}
C8_HD Dual operator/(double a, Dual const& b) {
  double const r = c8_rcp(b.v);
  double const q = a * r;
  return Dual(q, -q * b.d * r);
}
C8_HD Dual& operator+=(Dual& a, Dual const& b) { a.v += b.v; a.d += b.d; return a; }
C8_HD Dual& operator-=(Dual& a, Dual const& b) { a.v -= b.v; a.d -= b.d; return a; }
C8_HD Dual& operator+=(Dual& a, double b) { a.v += b; return a; }
C8_HD Dual& operator-=(Dual& a, double b) { a.v -= b; return a; }

C8_HD Dual c8_sqrt(Dual const& a) {
  double const s = sqrt(a.v);
  return Dual(s, a.d / (2. * s));
}
C8_HD double c8_sqrt(double a) { return sqrt(a); }
C8_HD Dual c8_cbrt(Dual const& a) {
  double const c = cbrt(a.v);
  return Dual(c, a.d / (3. * c * c));
}
C8_HD double c8_cbrt(double a) { return cbrt(a); }
C8_HD Dual c8_exp(Dual const& a) {
  double const e = exp(a.v);
  return Dual(e, e * a.d);
}
C8_HD double c8_exp(double a) { return exp(a); }
C8_HD Dual c8_log(Dual const& a) { return Dual(log(a.v), a.d / a.v); }
C8_HD double c8_log(double a) { return log(a); }
C8_HD Dual c8_cos(Dual const& a) { return Dual(cos(a.v), -sin(a.v) * a.d); }
C8_HD double c8_cos(double a) { return cos(a); }
C8_HD Dual c8_acos(Dual const& a) { return Dual(acos(a.v), -a.d / sqrt(1. - a.v * a.v)); }
C8_HD double c8_acos(double a) { return acos(a); }
C8_HD Dual c8_abs(Dual const& a) { return a.v >= 0. ? a : Dual(-a.v, -a.d); }
C8_HD double c8_abs(double a) { return fabs(a); }
// pow(a, b) with both arguments differentiable (Sacado's rule: zero derivative at a == 0).  Sacado drops the term of an
// operand that carries no derivatives (an unseeded parameter as exponent: no log of the base, so a negative base -- the
// hardening variable inside a Newton iteration -- gives a finite derivative); with one tangent per lane that is the
// operand whose tangent is zero.
C8_HD Dual c8_pow(Dual const& a, Dual const& b) {
  double const r = pow(a.v, b.v);
  double d = 0.;
  if (a.v != 0.) {
    if (b.d != 0.) d += b.d * log(a.v);
    if (a.d != 0.) d += b.v * a.d / a.v;
    d *= r;
  }
  return Dual(r, d);
}
C8_HD double c8_pow(double a, double b) { return pow(a, b); }

// ---------------------------------------------------------------------------
template <class T> struct Tens3 {
  T xx, xy, xz, yx, yy, yz, zx, zy, zz;
};

template <class T> C8_HD Tens3<T> eye3() {
  Tens3<T> r;
  r.xx = T(1.); r.xy = T(0.); r.xz = T(0.);
  r.yx = T(0.); r.yy = T(1.); r.yz = T(0.);
  r.zx = T(0.); r.zy = T(0.); r.zz = T(1.);
  return r;
}
template <class T> C8_HD Tens3<T> operator+(Tens3<T> const& A, Tens3<T> const& B) {
  Tens3<T> r;
  r.xx = A.xx + B.xx; r.xy = A.xy + B.xy; r.xz = A.xz + B.xz;
  r.yx = A.yx + B.yx; r.yy = A.yy + B.yy; r.yz = A.yz + B.yz;
  r.zx = A.zx + B.zx; r.zy = A.zy + B.zy; r.zz = A.zz + B.zz;
  return r;
}
template <class T> C8_HD Tens3<T> operator-(Tens3<T> const& A, Tens3<T> const& B) {
  Tens3<T> r;
  r.xx = A.xx - B.xx; r.xy = A.xy - B.xy; r.xz = A.xz - B.xz;
  r.yx = A.yx - B.yx; r.yy = A.yy - B.yy; r.yz = A.yz - B.yz;
  r.zx = A.zx - B.zx; r.zy = A.zy - B.zy; r.zz = A.zz - B.zz;
  return r;
}
template <class S, class T> C8_HD Tens3<T> scale(S const& s, Tens3<T> const& A) {
  Tens3<T> r;
  r.xx = s * A.xx; r.xy = s * A.xy; r.xz = s * A.xz;
  r.yx = s * A.yx; r.yy = s * A.yy; r.yz = s * A.yz;
  r.zx = s * A.zx; r.zy = s * A.zy; r.zz = s * A.zz;
  return r;
}
// a1 b1 + a2 b2 + a3 b3 as one chain of fused multiply-adds: 3 instructions for doubles, 3 + 6 for dual numbers (the
// operator form adds three separately rounded products: 3 + 8)
C8_HD double dot3(double a1, double b1, double a2, double b2, double a3, double b3) {
  return fma(a3, b3, fma(a2, b2, a1 * b1));
}
C8_HD Dual dot3(Dual const& a1, Dual const& b1, Dual const& a2, Dual const& b2, Dual const& a3, Dual const& b3) {
  return Dual(fma(a3.v, b3.v, fma(a2.v, b2.v, a1.v * b1.v)),
              fma(a3.d, b3.v, fma(a3.v, b3.d, fma(a2.d, b2.v, fma(a2.v, b2.d, fma(a1.d, b1.v, a1.v * b1.d))))));
}
template <class T> C8_HD Tens3<T> matmul(Tens3<T> const& A, Tens3<T> const& B) {
  Tens3<T> r;
  r.xx = dot3(A.xx, B.xx, A.xy, B.yx, A.xz, B.zx);
  r.xy = dot3(A.xx, B.xy, A.xy, B.yy, A.xz, B.zy);
  r.xz = dot3(A.xx, B.xz, A.xy, B.yz, A.xz, B.zz);
  r.yx = dot3(A.yx, B.xx, A.yy, B.yx, A.yz, B.zx);
  r.yy = dot3(A.yx, B.xy, A.yy, B.yy, A.yz, B.zy);
  r.yz = dot3(A.yx, B.xz, A.yy, B.yz, A.yz, B.zz);
  r.zx = dot3(A.zx, B.xx, A.zy, B.yx, A.zz, B.zx);
  r.zy = dot3(A.zx, B.xy, A.zy, B.yy, A.zz, B.zy);
  r.zz = dot3(A.zx, B.xz, A.zy, B.yz, A.zz, B.zz);
  return r;
}
template <class T> C8_HD Tens3<T> transpose(Tens3<T> const& A) {
  Tens3<T> r;
  r.xx = A.xx; r.xy = A.yx; r.xz = A.zx;
  r.yx = A.xy; r.yy = A.yy; r.yz = A.zy;
  r.zx = A.xz; r.zy = A.yz; r.zz = A.zz;
  return r;
}
template <class T> C8_HD T trace(Tens3<T> const& A) { return A.xx + A.yy + A.zz; }
// a b - c d as one chain: 2 instructions for doubles, 2 + 4 for dual numbers (the operator form: 2 + 5)
C8_HD double diff2(double a, double b, double c, double d) { return fma(a, b, -(c * d)); }
C8_HD Dual diff2(Dual const& a, Dual const& b, Dual const& c, Dual const& d) {
  return Dual(fma(a.v, b.v, -(c.v * d.v)), fma(a.v, b.d, fma(a.d, b.v, -fma(c.v, d.d, c.d * d.v))));
}
template <class T> C8_HD T det(Tens3<T> const& A) {
  return dot3(A.xx, diff2(A.yy, A.zz, A.yz, A.zy), A.xy, diff2(A.yz, A.zx, A.yx, A.zz), A.xz, diff2(A.yx, A.zy, A.yy, A.zx));
}
// cofactor matrix C with A^{-1} = C^T / det(A)  (mechanics.cpp:85-94 writes the same entries)
template <class T> C8_HD Tens3<T> cofactor(Tens3<T> const& F) {
  Tens3<T> C;
  C.xx = diff2(F.yy, F.zz, F.yz, F.zy);
  C.xy = diff2(F.yz, F.zx, F.yx, F.zz);
  C.xz = diff2(F.yx, F.zy, F.yy, F.zx);
  C.yx = diff2(F.xz, F.zy, F.xy, F.zz);
  C.yy = diff2(F.xx, F.zz, F.xz, F.zx);
  C.yz = diff2(F.xy, F.zx, F.xx, F.zy);
  C.zx = diff2(F.xy, F.yz, F.xz, F.yy);
  C.zy = diff2(F.xz, F.yx, F.xx, F.yz);
  C.zz = diff2(F.xx, F.yy, F.xy, F.yx);
  return C;
}
template <class T> C8_HD Tens3<T> inverse(Tens3<T> const& A) {
  Tens3<T> const C = cofactor(A);
  T const dt = dot3(A.xx, C.xx, A.xy, C.xy, A.xz, C.xz);
  T const r = 1. / dt;
  return scale(r, transpose(C));
}
template <class T> C8_HD Tens3<T> dev(Tens3<T> const& A) {
  T const th = trace(A) * (1. / 3.);
  Tens3<T> r = A;
  r.xx = A.xx - th; r.yy = A.yy - th; r.zz = A.zz - th;
  return r;
}
template <class T> C8_HD T norm(Tens3<T> const& A) {
  T const s = A.xx * A.xx + A.xy * A.xy + A.xz * A.xz + A.yx * A.yx + A.yy * A.yy + A.yz * A.yz +
              A.zx * A.zx + A.zy * A.zy + A.zz * A.zz;
  return c8_sqrt(s);
}
// ---------------------------------------------------------------------------
// minitensor::eig_spd_cos (Trilinos MiniTensor, third party): closed-form eigen-decomposition of a symmetric 3 x 3 tensor
// (Scherzinger & Dohrmann, CMAME 197 (2008) 4007-4015), restated from the published algorithm with the scalar type as a
// template parameter, so that derivatives flow through it as in the reference: the most distinct eigenvalue D[2] from the
// trigonometric solution of the characteristic equation of the deviator, its eigenvector from the column space of
// (A' - D[2] I) by Gram-Schmidt with column pivoting, the other two from the 2 x 2 problem on the orthogonal complement.
// An (almost) diagonal tensor returns V = I, D = diag(A) at once.  Eigenvectors are the COLUMNS of V.  Column choices
// are selects, not indexed stores, so everything stays in registers.
// ---------------------------------------------------------------------------
template <class T> C8_HD T sel3(int k, T const& a, T const& b, T const& c) { return k == 0 ? a : (k == 1 ? b : c); }
template <class T> C8_HD void eig_spd_cos(Tens3<T> const& A, Tens3<T>& V, T* D) {
  double const off = val(A.xy) * val(A.xy) + val(A.xz) * val(A.xz) + val(A.yx) * val(A.yx) + val(A.yz) * val(A.yz) +
                     val(A.zx) * val(A.zx) + val(A.zy) * val(A.zy);
  if (sqrt(off) <= 2.220446049250313e-16) {
    V = eye3<T>();
    D[0] = A.xx; D[1] = A.yy; D[2] = A.zz;
    return;
  }
  double const pi = 3.141592653589793238;
  T const trA = (1. / 3.) * trace(A);
  Tens3<T> Ap = A;
  Ap.xx = A.xx - trA; Ap.yy = A.yy - trA; Ap.zz = A.zz - trA;
  T const J2 = -(Ap.xx * Ap.yy + Ap.yy * Ap.zz + Ap.zz * Ap.xx) + Ap.xy * Ap.xy + Ap.yz * Ap.yz + Ap.zx * Ap.zx;
  T const J3 = det(Ap);
  if (val(J2) <= 1.e-30) {  // volumetric tensor
    D[0] = trA; D[1] = trA; D[2] = trA;
    V = eye3<T>();
    return;
  }
  T const t1 = 3. / J2;
  T const rhs = (J3 / 2.) * c8_sqrt(t1 * t1 * t1);
  T theta = T(pi / 2. * (1. - (val(rhs) < 0. ? -1. : 1.)));
  if (fabs(val(rhs)) <= 1.) theta = c8_acos(rhs);
  T thetad3 = theta / 3.;
  if (val(thetad3) > pi / 6.) thetad3 = thetad3 + 2. * pi / 3.;
  T const lam = 2. * c8_cos(thetad3) * c8_sqrt(J2 / 3.);
  // columns of R = A' - lam I
  T c0[3] = {Ap.xx - lam, Ap.yx, Ap.zx}, c1[3] = {Ap.xy, Ap.yy - lam, Ap.zy}, c2[3] = {Ap.xz, Ap.yz, Ap.zz - lam};
  T const a0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
  T const a1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
  T const a2 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
  int k = 0;
  if (val(a1) > val(a0)) k = 1;
  if (val(a2) > val(sel3(k, a0, a1, a2))) k = 2;
  T const nk = c8_sqrt(sel3(k, a0, a1, a2));
  T s1[3], ca[3], cb[3];  // the dominant column, normalised, and the other two in the order (k+1, k+2) mod 3
  C8_UNROLL
  for (int i = 0; i < 3; ++i) {
    s1[i] = sel3(k, c0[i], c1[i], c2[i]) / nk;
    ca[i] = sel3(k, c1[i], c2[i], c0[i]);
    cb[i] = sel3(k, c2[i], c0[i], c1[i]);
  }
  T const d0 = s1[0] * ca[0] + s1[1] * ca[1] + s1[2] * ca[2];
  T const d1 = s1[0] * cb[0] + s1[1] * cb[1] + s1[2] * cb[2];
  C8_UNROLL
  for (int i = 0; i < 3; ++i) { ca[i] = ca[i] - d0 * s1[i]; cb[i] = cb[i] - d1 * s1[i]; }
  T const b0 = ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2];
  T const b1 = cb[0] * cb[0] + cb[1] * cb[1] + cb[2] * cb[2];
  bool const p = fabs(val(b1)) > fabs(val(b0));
  T const nk2 = c8_sqrt(p ? b1 : b0);
  T s2[3];
  C8_UNROLL
  for (int i = 0; i < 3; ++i) s2[i] = (p ? cb[i] : ca[i]) / nk2;
  // eigenvector of lam: s1 x s2
  T v2[3] = {s1[1] * s2[2] - s1[2] * s2[1], s1[2] * s2[0] - s1[0] * s2[2], s1[0] * s2[1] - s1[1] * s2[0]};
  T mag = c8_sqrt(v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2]);
  C8_UNROLL
  for (int i = 0; i < 3; ++i) v2[i] = v2[i] / mag;
  // the 2 x 2 problem on span{s1, s2}
  T ak[3], ak2[3];
  ak[0] = Ap.xx * s1[0] + Ap.xy * s1[1] + Ap.xz * s1[2];
  ak[1] = Ap.yx * s1[0] + Ap.yy * s1[1] + Ap.yz * s1[2];
  ak[2] = Ap.zx * s1[0] + Ap.zy * s1[1] + Ap.zz * s1[2];
  ak2[0] = Ap.xx * s2[0] + Ap.xy * s2[1] + Ap.xz * s2[2];
  ak2[1] = Ap.yx * s2[0] + Ap.yy * s2[1] + Ap.yz * s2[2];
  ak2[2] = Ap.zx * s2[0] + Ap.zy * s2[1] + Ap.zz * s2[2];
  T rm00 = s1[0] * ak[0] + s1[1] * ak[1] + s1[2] * ak[2];
  T const rm01 = s1[0] * ak2[0] + s1[1] * ak2[1] + s1[2] * ak2[2];
  T rm11 = s2[0] * ak2[0] + s2[1] * ak2[1] + s2[2] * ak2[2];
  T const b = 0.5 * (rm00 - rm11);
  double const fac = val(b) < 0. ? -1. : 1.;
  T const arg = b * b + rm01 * rm01;
  T lam0;
  if (val(arg) == 0.) lam0 = rm11 + b;
  else lam0 = rm11 + b - fac * c8_sqrt(arg);
  T const lam1 = rm00 + rm11 - lam0;
  rm00 = rm00 - lam0;
  rm11 = rm11 - lam0;
  T const q0 = rm00 * rm00 + rm01 * rm01, q1 = rm01 * rm01 + rm11 * rm11;
  bool const k3 = val(q1) > val(q0);
  T m0 = k3 ? rm01 : rm00, m1 = k3 ? rm11 : rm01;
  if (val(k3 ? q1 : q0) == 0.) { m0 = T(1.); m1 = T(0.); }
  T v0[3];
  C8_UNROLL
  for (int i = 0; i < 3; ++i) v0[i] = m0 * s2[i] - m1 * s1[i];
  mag = c8_sqrt(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]);
  C8_UNROLL
  for (int i = 0; i < 3; ++i) v0[i] = v0[i] / mag;
  T v1[3] = {v0[1] * v2[2] - v0[2] * v2[1], v0[2] * v2[0] - v0[0] * v2[2], v0[0] * v2[1] - v0[1] * v2[0]};
  mag = c8_sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
  C8_UNROLL
  for (int i = 0; i < 3; ++i) v1[i] = v1[i] / mag;
  V.xx = v0[0]; V.yx = v0[1]; V.zx = v0[2];
  V.xy = v1[0]; V.yy = v1[1]; V.zy = v1[2];
  V.xz = v2[0]; V.yz = v2[1]; V.zz = v2[2];
  D[0] = lam0 + trA; D[1] = lam1 + trA; D[2] = lam + trA;
}
// w (V e_c)(V e_c)^T added to S: the weighted eigen-dyad of column c
template <class T> C8_HD void add_dyad_col(Tens3<T>& S, T const& w, Tens3<T> const& V, int c) {
  T const x = sel3(c, V.xx, V.xy, V.xz), y = sel3(c, V.yx, V.yy, V.yz), z = sel3(c, V.zx, V.zy, V.zz);
  S.xx = S.xx + w * (x * x); S.xy = S.xy + w * (x * y); S.xz = S.xz + w * (x * z);
  S.yx = S.yx + w * (y * x); S.yy = S.yy + w * (y * y); S.yz = S.yz + w * (y * z);
  S.zx = S.zx + w * (z * x); S.zy = S.zy + w * (z * y); S.zz = S.zz + w * (z * z);
}

// minitensor::polar_rotation (Trilinos MiniTensor, third party): the rotation R of F = R U by Newton's iteration
// X <- (mu X + X^-T / mu) / 2 with Higham's 1-norm / infinity-norm scaling, differentiated through like any other
// arithmetic (global_residual.hpp:302-305 calls it on the FAD deformation gradient).  The trip count depends on
// values only, so it is the same for every derivative slot of a point.
template <class T> C8_HD T norm_1(Tens3<T> const& A) {  // largest absolute column sum
  T const c0 = c8_abs(A.xx) + c8_abs(A.yx) + c8_abs(A.zx);
  T const c1 = c8_abs(A.xy) + c8_abs(A.yy) + c8_abs(A.zy);
  T const c2 = c8_abs(A.xz) + c8_abs(A.yz) + c8_abs(A.zz);
  T best = c0;
  if (val(c1) > val(best)) best = c1;
  if (val(c2) > val(best)) best = c2;
  return best;
}
template <class T> C8_HD T norm_infinity(Tens3<T> const& A) {  // largest absolute row sum
  T const r0 = c8_abs(A.xx) + c8_abs(A.xy) + c8_abs(A.xz);
  T const r1 = c8_abs(A.yx) + c8_abs(A.yy) + c8_abs(A.yz);
  T const r2 = c8_abs(A.zx) + c8_abs(A.zy) + c8_abs(A.zz);
  T best = r0;
  if (val(r1) > val(best)) best = r1;
  if (val(r2) > val(best)) best = r2;
  return best;
}
template <class T> C8_HD double norm_val(Tens3<T> const& A) {  // value of the Frobenius norm
  double const s = val(A.xx) * val(A.xx) + val(A.xy) * val(A.xy) + val(A.xz) * val(A.xz) + val(A.yx) * val(A.yx) +
                   val(A.yy) * val(A.yy) + val(A.yz) * val(A.yz) + val(A.zx) * val(A.zx) + val(A.zy) * val(A.zy) +
                   val(A.zz) * val(A.zz);
  return sqrt(s);
}
template <class T> C8_HD Tens3<T> polar_rotation(Tens3<T> const& A) {
  bool scaling = true;
  double const tol_scale = 0.01;
  double const sqrt_tol_conv = 1.9611031010039037e-08;  // sqrt(sqrt(3) * machine epsilon)
  Tens3<T> X = A;
  double gamma = 2.0;
  C8_NOUNROLL
  for (int num_iter = 0; num_iter < 128; ++num_iter) {
    Tens3<T> const Y = inverse(X);
    T mu = T(1.0);
    if (scaling) {
      mu = (norm_1(Y) * norm_infinity(Y)) / (norm_1(X) * norm_infinity(X));
      mu = c8_sqrt(c8_sqrt(mu));
    }
    Tens3<T> const Z = scale(0.5, scale(mu, X) + scale(1. / mu, transpose(Y)));
    Tens3<T> const D = Z - X;
    double const nD = norm_val(D);
    double const delta = nD / norm_val(Z);
    if (scaling && delta < tol_scale) scaling = false;
    bool const end_iter = nD <= sqrt_tol_conv || (delta > 0.5 * gamma && !scaling);
    X = Z;
    gamma = delta;
    if (end_iter) break;
  }
  return X;
}

// symmetric tensor from the packed local-variable order (00,01,02,11,12,22)
// (local_residual.cpp:206-216)
template <class T> C8_HD Tens3<T> sym6(T const* s) {
  Tens3<T> r;
  r.xx = s[0]; r.xy = s[1]; r.xz = s[2];
  r.yx = s[1]; r.yy = s[3]; r.yz = s[4];
  r.zx = s[2]; r.zy = s[4]; r.zz = s[5];
  return r;
}
template <class T> C8_HD void pack_sym6(Tens3<T> const& t, T* s) {  // local_residual.cpp:572-577
  s[0] = t.xx; s[1] = t.xy; s[2] = t.xz; s[3] = t.yy; s[4] = t.yz; s[5] = t.zz;
}
// The same in DIM dimensions.  The reference's tensors have a run-time dimension (MiniTensor): on a 2-D mesh every
// model works on 2 x 2 tensors and packs symmetric ones as (00,01,11) (local_residual.cpp:197-204, :565-570).  Here a
// 2 x 2 tensor is a Tens3 whose out-of-plane entries are zero -- sums over nine entries then equal the sums over four.
template <int DIM, class T> C8_HD Tens3<T> sym_dim(T const* s) {
  if (DIM == 3) return sym6(s);
  Tens3<T> r;
  r.xx = s[0]; r.xy = s[1]; r.xz = T(0.);
  r.yx = s[1]; r.yy = s[2]; r.yz = T(0.);
  r.zx = T(0.); r.zy = T(0.); r.zz = T(0.);
  return r;
}
template <int DIM, class T> C8_HD void pack_sym_dim(Tens3<T> const& t, T* s) {
  if (DIM == 3) { pack_sym6(t, s); return; }
  s[0] = t.xx; s[1] = t.xy; s[2] = t.yy;
}
// A - s I with I the DIM x DIM identity
template <int DIM, class T, class S> C8_HD Tens3<T> minus_s_eye(Tens3<T> const& A, S const& s) {
  Tens3<T> r = A;
  r.xx = A.xx - s; r.yy = A.yy - s;
  if (DIM == 3) r.zz = A.zz - s;
  return r;
}

}  // namespace c8
