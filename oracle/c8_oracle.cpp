// oracle/c8_oracle.cpp
//
// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
//
// CPU restatement of the CALIBR8 per-element residual / Jacobian assembly and
// adjoint-sensitivity path, written from the reference's algorithm description
// (file:line citations are into /root/reference/source/calibr8/src/).  It is the
// parity checker for the HIP path in calibr8_amd/ and the timed "port" CPU
// baseline of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product never does.
//
// Pinning status: the reference binary cannot be built here (Trilinos, SCOREC
// and Eigen are absent), and no reference test pins element residual/Jacobian
// entries.  This restatement is pinned by the reference's own end-to-end
// regression values on the shipped cube mesh (tests/test_oracle_pins.py:
// cube_elastic 5.0e-3 @1e-6, cube_hyper_J2, cube_hyperelasticity[_traction]
// @1e-4) and by finite-difference checks in the reference's own style
// (main_inverse.cpp:126-158).  At the 1e-12 level it is "parity unpinned"
// against the reference binary; see DESIGN.md.
//
// Third-party arithmetic restated from published behaviour (sources are not in
// /root/reference): Sacado::Fad::SLFad (Trilinos @33f2129), MiniTensor 3x3
// algebra, Eigen::FullPivLU (Eigen @bc3b398), apf Lagrange shapes and Gauss
// rules (SCOREC core @483760e).  hex8 is an extension: the reference only
// runs simplices (disc.cpp:165).
//
// The per-quadrature-point sequence of AD passes, local Newton solve, dxi/dx
// condensation and per-point scatter follows evaluations.cpp:12-154 exactly;
// nothing is restructured for speed.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <type_traits>
#include <set>
#include <vector>

namespace c8o {

// ---------------------------------------------------------------------------
// Forward AD scalar: static storage, run-time size (defines.hpp:23-26,
// Sacado::Fad::SLFad<double,16>; 32 slots here so hex8 mixed elements fit).
// A value of size 0 is a constant.  Derivative formulas follow Sacado's
// expression-template rules term for term.
// ---------------------------------------------------------------------------
constexpr int NMAX = 32;

struct Fad {
  double v;
  int n;
  double d[NMAX];
  Fad() : v(0.), n(0) {}
  Fad(double x) : v(x), n(0) {}
  double val() const { return v; }
  double dx(int i) const { return i < n ? d[i] : 0.; }
  void diff(int i, int nn) {
    n = nn;
    for (int k = 0; k < nn; ++k) d[k] = 0.;
    d[i] = 1.;
  }
  Fad& operator=(double x) {
    v = x;
    n = 0;
    return *this;
  }
};

inline double val(double x) { return x; }
inline double val(Fad const& x) { return x.v; }

inline Fad operator-(Fad const& a) {
  Fad r;
  r.v = -a.v;
  r.n = a.n;
  for (int i = 0; i < a.n; ++i) r.d[i] = -a.d[i];
  return r;
}
inline Fad operator+(Fad const& a, Fad const& b) {
  Fad r;
  r.v = a.v + b.v;
  r.n = std::max(a.n, b.n);
  if (a.n && b.n) for (int i = 0; i < r.n; ++i) r.d[i] = a.d[i] + b.d[i];
  else if (a.n) for (int i = 0; i < r.n; ++i) r.d[i] = a.d[i];
  else for (int i = 0; i < r.n; ++i) r.d[i] = b.d[i];
  return r;
}
inline Fad operator-(Fad const& a, Fad const& b) {
  Fad r;
  r.v = a.v - b.v;
  r.n = std::max(a.n, b.n);
  if (a.n && b.n) for (int i = 0; i < r.n; ++i) r.d[i] = a.d[i] - b.d[i];
  else if (a.n) for (int i = 0; i < r.n; ++i) r.d[i] = a.d[i];
  else for (int i = 0; i < r.n; ++i) r.d[i] = -b.d[i];
  return r;
}
inline Fad operator*(Fad const& a, Fad const& b) {
  Fad r;
  r.v = a.v * b.v;
  r.n = std::max(a.n, b.n);
  if (a.n && b.n) for (int i = 0; i < r.n; ++i) r.d[i] = a.v * b.d[i] + a.d[i] * b.v;
  else if (a.n) for (int i = 0; i < r.n; ++i) r.d[i] = a.d[i] * b.v;
  else for (int i = 0; i < r.n; ++i) r.d[i] = a.v * b.d[i];
  return r;
}
inline Fad operator/(Fad const& a, Fad const& b) {
  Fad r;
  r.v = a.v / b.v;
  r.n = std::max(a.n, b.n);
  double const b2 = b.v * b.v;
  if (a.n && b.n) for (int i = 0; i < r.n; ++i) r.d[i] = (a.d[i] * b.v - a.v * b.d[i]) / b2;
  else if (a.n) for (int i = 0; i < r.n; ++i) r.d[i] = a.d[i] / b.v;
  else for (int i = 0; i < r.n; ++i) r.d[i] = -a.v * b.d[i] / b2;
  return r;
}
inline Fad operator+(Fad const& a, double b) { Fad r = a; r.v = a.v + b; return r; }
inline Fad operator+(double a, Fad const& b) { Fad r = b; r.v = a + b.v; return r; }
inline Fad operator-(Fad const& a, double b) { Fad r = a; r.v = a.v - b; return r; }
inline Fad operator-(double a, Fad const& b) { Fad r = -b; r.v = a - b.v; return r; }
inline Fad operator*(Fad const& a, double b) {
  Fad r; r.v = a.v * b; r.n = a.n;
  for (int i = 0; i < a.n; ++i) r.d[i] = a.d[i] * b;
  return r;
}
inline Fad operator*(double a, Fad const& b) {
  Fad r; r.v = a * b.v; r.n = b.n;
  for (int i = 0; i < b.n; ++i) r.d[i] = a * b.d[i];
  return r;
}
inline Fad operator/(Fad const& a, double b) {
  Fad r; r.v = a.v / b; r.n = a.n;
  for (int i = 0; i < a.n; ++i) r.d[i] = a.d[i] / b;
  return r;
}
inline Fad operator/(double a, Fad const& b) {
  Fad r; r.v = a / b.v; r.n = b.n;
  double const b2 = b.v * b.v;
  for (int i = 0; i < b.n; ++i) r.d[i] = -a * b.d[i] / b2;
  return r;
}
inline Fad& operator+=(Fad& a, Fad const& b) { a = a + b; return a; }
inline Fad& operator-=(Fad& a, Fad const& b) { a = a - b; return a; }
inline Fad& operator*=(Fad& a, Fad const& b) { a = a * b; return a; }
inline Fad& operator/=(Fad& a, Fad const& b) { a = a / b; return a; }
inline Fad& operator+=(Fad& a, double b) { a.v += b; return a; }
inline Fad& operator-=(Fad& a, double b) { a.v -= b; return a; }
inline Fad& operator/=(Fad& a, double b) { a = a / b; return a; }
inline bool operator>(Fad const& a, double b) { return a.v > b; }
inline bool operator<(Fad const& a, double b) { return a.v < b; }

inline Fad sqrt(Fad const& a) {
  Fad r; r.v = std::sqrt(a.v); r.n = a.n;
  double const s = 2. * r.v;
  for (int i = 0; i < a.n; ++i) r.d[i] = a.d[i] / s;
  return r;
}
inline Fad cbrt(Fad const& a) {
  Fad r; r.v = std::cbrt(a.v); r.n = a.n;
  double const s = 3. * std::cbrt(a.v * a.v);
  for (int i = 0; i < a.n; ++i) r.d[i] = a.d[i] / s;
  return r;
}
inline Fad exp(Fad const& a) {
  Fad r; r.v = std::exp(a.v); r.n = a.n;
  for (int i = 0; i < a.n; ++i) r.d[i] = r.v * a.d[i];
  return r;
}
inline Fad log(Fad const& a) {
  Fad r; r.v = std::log(a.v); r.n = a.n;
  for (int i = 0; i < a.n; ++i) r.d[i] = a.d[i] / a.v;
  return r;
}
inline Fad cos(Fad const& a) {
  Fad r; r.v = std::cos(a.v); r.n = a.n;
  double const s = -std::sin(a.v);
  for (int i = 0; i < a.n; ++i) r.d[i] = s * a.d[i];
  return r;
}
inline Fad acos(Fad const& a) {
  Fad r; r.v = std::acos(a.v); r.n = a.n;
  double const s = -1. / std::sqrt(1. - a.v * a.v);
  for (int i = 0; i < a.n; ++i) r.d[i] = s * a.d[i];
  return r;
}
inline Fad abs(Fad const& a) {
  Fad r; r.v = std::fabs(a.v); r.n = a.n;
  double const s = a.v >= 0. ? 1. : -1.;
  for (int i = 0; i < a.n; ++i) r.d[i] = s * a.d[i];
  return r;
}
inline Fad pow(Fad const& a, Fad const& b) {  // Sacado's PowerOp: which formula applies depends on which operands carry derivatives
  Fad r; r.v = std::pow(a.v, b.v); r.n = std::max(a.n, b.n);
  for (int i = 0; i < r.n; ++i) {
    if (a.v == 0.) { r.d[i] = 0.; continue; }
    if (a.n > 0 && b.n > 0) r.d[i] = (b.dx(i) * std::log(a.v) + b.v * a.dx(i) / a.v) * r.v;
    else if (a.n > 0) r.d[i] = b.v * a.dx(i) / a.v * r.v;       // constant exponent (unseeded parameter): no log of the base
    else r.d[i] = b.dx(i) * std::log(a.v) * r.v;                // constant base
  }
  return r;
}
using std::abs;
using std::acos;
using std::cos;
using std::log;
using std::cbrt;
using std::exp;
using std::pow;
using std::sqrt;

// ---------------------------------------------------------------------------
// Tensor algebra (the MiniTensor subset the hot models use, a15 of SURVEY.md section 8a):
// norm (Frobenius), trace, transpose, eye, det, inverse, dev.  MiniTensor tensors carry a
// RUN-TIME dimension (defines.hpp:31-37): 3 in 3-D, 2 in the 2-D decks (tri3), where every
// model runs on 2 x 2 tensors (small_J2.cpp:187 `eye<T>(ndims)`).  Here a tensor is stored
// 3 x 3 with `dim`; the entries outside dim x dim are zero and stay zero under every
// operation below, so sums over all nine entries equal the sums over dim x dim bit for bit.
// ---------------------------------------------------------------------------
template <class T> struct Tens {
  T a[3][3] = {};
  int dim = 3;
  T& operator()(int i, int j) { return a[i][j]; }
  T const& operator()(int i, int j) const { return a[i][j]; }
};
template <class T> struct Vec {
  T a[3] = {};
  T& operator()(int i) { return a[i]; }
  T const& operator()(int i) const { return a[i]; }
};
template <class S> struct is_tens : std::false_type {};
template <class T> struct is_tens<Tens<T>> : std::true_type {};

template <class T> Tens<T> eye(int dim = 3) {
  Tens<T> r;
  r.dim = dim;
  for (int i = 0; i < dim; ++i) for (int j = 0; j < dim; ++j) r(i, j) = (i == j) ? 1. : 0.;
  return r;
}
template <class T> Tens<T> operator+(Tens<T> const& A, Tens<T> const& B) {
  Tens<T> r;
  r.dim = A.dim;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = A(i, j) + B(i, j);
  return r;
}
template <class T> Tens<T> operator-(Tens<T> const& A, Tens<T> const& B) {
  Tens<T> r;
  r.dim = A.dim;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = A(i, j) - B(i, j);
  return r;
}
template <class T> Tens<T> operator*(Tens<T> const& A, Tens<T> const& B) {
  Tens<T> r;
  r.dim = A.dim;
  for (int i = 0; i < A.dim; ++i) for (int j = 0; j < A.dim; ++j) {
    T s = A(i, 0) * B(0, j);
    for (int k = 1; k < A.dim; ++k) s += A(i, k) * B(k, j);
    r(i, j) = s;
  }
  return r;
}
template <class S, class T, typename std::enable_if<!is_tens<S>::value, int>::type = 0>
Tens<T> operator*(S const& s, Tens<T> const& A) {
  Tens<T> r;
  r.dim = A.dim;
  for (int i = 0; i < A.dim; ++i) for (int j = 0; j < A.dim; ++j) r(i, j) = s * A(i, j);
  return r;
}
template <class S, class T, typename std::enable_if<!is_tens<S>::value, int>::type = 0>
Tens<T> operator*(Tens<T> const& A, S const& s) {
  Tens<T> r;
  r.dim = A.dim;
  for (int i = 0; i < A.dim; ++i) for (int j = 0; j < A.dim; ++j) r(i, j) = A(i, j) * s;
  return r;
}
template <class S, class T> Tens<T> operator/(Tens<T> const& A, S const& s) {
  Tens<T> r;
  r.dim = A.dim;
  for (int i = 0; i < A.dim; ++i) for (int j = 0; j < A.dim; ++j) r(i, j) = A(i, j) / s;
  return r;
}
template <class T> Tens<T> transpose(Tens<T> const& A) {
  Tens<T> r;
  r.dim = A.dim;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = A(j, i);
  return r;
}
template <class T> T trace(Tens<T> const& A) { return A(0, 0) + A(1, 1) + A(2, 2); }
template <class T> T det(Tens<T> const& A) {
  if (A.dim == 2) return A(0, 0) * A(1, 1) - A(1, 0) * A(0, 1);
  return -A(0, 2) * A(1, 1) * A(2, 0) + A(0, 1) * A(1, 2) * A(2, 0) +
         A(0, 2) * A(1, 0) * A(2, 1) - A(0, 0) * A(1, 2) * A(2, 1) -
         A(0, 1) * A(1, 0) * A(2, 2) + A(0, 0) * A(1, 1) * A(2, 2);
}
template <class T> Tens<T> inverse(Tens<T> const& A) {
  T const dt = det(A);
  Tens<T> B;
  B.dim = A.dim;
  if (A.dim == 2) {
    B(0, 0) = A(1, 1) / dt; B(0, 1) = -A(0, 1) / dt;
    B(1, 0) = -A(1, 0) / dt; B(1, 1) = A(0, 0) / dt;
    return B;
  }
  B(0, 0) = -A(1, 2) * A(2, 1) + A(1, 1) * A(2, 2);
  B(0, 1) = A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2);
  B(0, 2) = -A(0, 2) * A(1, 1) + A(0, 1) * A(1, 2);
  B(1, 0) = A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2);
  B(1, 1) = -A(0, 2) * A(2, 0) + A(0, 0) * A(2, 2);
  B(1, 2) = A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2);
  B(2, 0) = -A(1, 1) * A(2, 0) + A(1, 0) * A(2, 1);
  B(2, 1) = A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1);
  B(2, 2) = -A(0, 1) * A(1, 0) + A(0, 0) * A(1, 1);
  return B / dt;
}
template <class T> Tens<T> dev(Tens<T> const& A) {
  T const th = trace(A) / 3.;  // only 3-D models call this
  return A - th * eye<T>(A.dim);
}
template <class T> T norm(Tens<T> const& A) {
  T s = A(0, 0) * A(0, 0);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (i || j) s += A(i, j) * A(i, j);
  return sqrt(s);
}

// ---------------------------------------------------------------------------
// minitensor::eig_spd_cos (Trilinos MiniTensor, MiniTensor_LinearAlgebra.t.h; third party, not in /root/reference --
// the reference pins Trilinos through its build): closed-form eigen-decomposition of a symmetric
// 3 x 3 tensor (Scherzinger & Dohrmann, CMAME 197 (2008) 4007-4015), restated from the published algorithm: the most
// distinct eigenvalue from the trigonometric solution of the characteristic equation of the deviator, its eigenvector
// from the column space of (A' - lambda I) by Gram-Schmidt with column pivoting, the other two from the 2 x 2 problem on
// the orthogonal complement.  Everything is templated on the scalar, so derivatives flow through it as in the
// reference.  An (almost) diagonal input returns V = I, D = A at once.  Returned: eigenvectors in the COLUMNS of V.
// The yield functions that call it are symmetric in the eigenpairs, so neither their order nor the signs of the
// vectors matter to the callers.
// ---------------------------------------------------------------------------
template <class T> void eig_spd_cos(Tens<T> const& A, Tens<T>& V, Tens<T>& D) {
  V = Tens<T>();
  D = Tens<T>();
  double off = 0.;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (i != j) off += val(A(i, j)) * val(A(i, j));
  if (std::sqrt(off) <= std::numeric_limits<double>::epsilon()) {
    V = eye<T>();
    D = A;
    return;
  }
  double const pi = std::acos(-1.);
  int const ii[3][2] = {{1, 2}, {2, 0}, {0, 1}};
  T const trA = (1. / 3.) * trace(A);
  Tens<T> const Ap = A - trA * eye<T>();
  T const J2 = -(Ap(0, 0) * Ap(1, 1) + Ap(1, 1) * Ap(2, 2) + Ap(2, 2) * Ap(0, 0)) + Ap(0, 1) * Ap(0, 1) + Ap(1, 2) * Ap(1, 2) +
               Ap(2, 0) * Ap(2, 0);  // second invariant of the deviator, = 1/2 Ap : Ap >= 0
  T const J3 = det(Ap);
  if (val(J2) <= 1.e-30) {  // volumetric tensor
    D(0, 0) = trA; D(1, 1) = trA; D(2, 2) = trA;
    V = eye<T>();
    return;
  }
  // cos(3 theta) = J3/2 (3/J2)^(3/2): the eigenvalue with the largest distance from the other two
  T const t1 = 3. / J2;
  T const rhs = (J3 / 2.) * sqrt(t1 * t1 * t1);
  T theta = T(pi / 2. * (1. - (val(rhs) < 0. ? -1. : 1.)));
  if (std::fabs(val(rhs)) <= 1.) theta = acos(rhs);
  T thetad3 = theta / 3.;
  if (val(thetad3) > pi / 6.) thetad3 = thetad3 + 2. * pi / 3.;
  D(2, 2) = 2. * cos(thetad3) * sqrt(J2 / 3.);
  Tens<T> R = Ap - D(2, 2) * eye<T>();
  // Gram-Schmidt with column pivoting on R: its two-dimensional column space is orthogonal to the eigenvector
  T a[3];
  for (int j = 0; j < 3; ++j) a[j] = R(0, j) * R(0, j) + R(1, j) * R(1, j) + R(2, j) * R(2, j);
  int k = 0;
  if (val(a[1]) > val(a[k])) k = 1;
  if (val(a[2]) > val(a[k])) k = 2;
  T const nk = sqrt(a[k]);
  for (int i = 0; i < 3; ++i) R(i, k) = R(i, k) / nk;
  T d0 = 0., d1 = 0.;
  for (int i = 0; i < 3; ++i) { d0 += R(i, k) * R(i, ii[k][0]); d1 += R(i, k) * R(i, ii[k][1]); }
  for (int i = 0; i < 3; ++i) { R(i, ii[k][0]) = R(i, ii[k][0]) - d0 * R(i, k); R(i, ii[k][1]) = R(i, ii[k][1]) - d1 * R(i, k); }
  T b0 = 0., b1 = 0.;
  for (int i = 0; i < 3; ++i) { b0 += R(i, ii[k][0]) * R(i, ii[k][0]); b1 += R(i, ii[k][1]) * R(i, ii[k][1]); }
  int const p = (std::fabs(val(b1)) > std::fabs(val(b0))) ? 1 : 0;
  int const k2 = ii[k][p];
  T const nk2 = sqrt(p ? b1 : b0);
  for (int i = 0; i < 3; ++i) R(i, k2) = R(i, k2) / nk2;
  // eigenvector of D(2,2): s1 x s2
  V(0, 2) = R(1, k) * R(2, k2) - R(2, k) * R(1, k2);
  V(1, 2) = R(2, k) * R(0, k2) - R(0, k) * R(2, k2);
  V(2, 2) = R(0, k) * R(1, k2) - R(1, k) * R(0, k2);
  T mag = sqrt(V(0, 2) * V(0, 2) + V(1, 2) * V(1, 2) + V(2, 2) * V(2, 2));
  for (int i = 0; i < 3; ++i) V(i, 2) = V(i, 2) / mag;
  // the 2 x 2 problem on span{s1, s2}
  T rk[3], rk2[3], ak[3], ak2[3];
  for (int i = 0; i < 3; ++i) { rk[i] = R(i, k); rk2[i] = R(i, k2); }
  for (int i = 0; i < 3; ++i) {
    ak[i] = Ap(i, 0) * rk[0] + Ap(i, 1) * rk[1] + Ap(i, 2) * rk[2];
    ak2[i] = Ap(i, 0) * rk2[0] + Ap(i, 1) * rk2[1] + Ap(i, 2) * rk2[2];
  }
  T rm00 = rk[0] * ak[0] + rk[1] * ak[1] + rk[2] * ak[2];
  T const rm01 = rk[0] * ak2[0] + rk[1] * ak2[1] + rk[2] * ak2[2];
  T rm11 = rk2[0] * ak2[0] + rk2[1] * ak2[1] + rk2[2] * ak2[2];
  T const b = 0.5 * (rm00 - rm11);
  double const fac = val(b) < 0. ? -1. : 1.;
  T const arg = b * b + rm01 * rm01;
  if (val(arg) == 0.) D(0, 0) = rm11 + b;
  else D(0, 0) = rm11 + b - fac * sqrt(arg);
  D(1, 1) = rm00 + rm11 - D(0, 0);
  rm00 = rm00 - D(0, 0);
  rm11 = rm11 - D(0, 0);
  T const c0 = rm00 * rm00 + rm01 * rm01, c1 = rm01 * rm01 + rm11 * rm11;
  int const k3 = (val(c1) > val(c0)) ? 1 : 0;
  T m0 = k3 ? rm01 : rm00, m1 = k3 ? rm11 : rm01;  // column k3 of the shifted 2 x 2 matrix
  if (val(k3 ? c1 : c0) == 0.) { m0 = 1.; m1 = 0.; }
  // eigenvector of D(0,0): orthogonal to that column within the plane
  for (int i = 0; i < 3; ++i) V(i, 0) = m0 * rk2[i] - m1 * rk[i];
  mag = sqrt(V(0, 0) * V(0, 0) + V(1, 0) * V(1, 0) + V(2, 0) * V(2, 0));
  for (int i = 0; i < 3; ++i) V(i, 0) = V(i, 0) / mag;
  // the last one completes the triad
  V(0, 1) = V(1, 0) * V(2, 2) - V(2, 0) * V(1, 2);
  V(1, 1) = V(2, 0) * V(0, 2) - V(0, 0) * V(2, 2);
  V(2, 1) = V(0, 0) * V(1, 2) - V(1, 0) * V(0, 2);
  mag = sqrt(V(0, 1) * V(0, 1) + V(1, 1) * V(1, 1) + V(2, 1) * V(2, 1));
  for (int i = 0; i < 3; ++i) V(i, 1) = V(i, 1) / mag;
  for (int i = 0; i < 3; ++i) D(i, i) = D(i, i) + trA;
}
template <class T> Tens<T> dyad_col(Tens<T> const& V, int c) {  // dyad(col(V, c), col(V, c))
  Tens<T> t;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t(i, j) = V(i, c) * V(j, c);
  return t;
}

// line_search.hpp:56-135: backtracking Armijo search with two-point cubic interpolation (shared by the global Newton
// loop, restated for it in tests/fe_driver.py, and the local solves of the Hosford / Barlat models)
struct LineSearchParams { double c1 = 1.e-4, backtrack_min = 0.5, backtrack_max = 0.9; int max_evals = 4; };
static double ls_cubic_min(double phi_0, double dphi_0, double a, double phi, double slope_a) {
  double const d1 = dphi_0 + slope_a - 3. * (phi_0 - phi) / (0. - a);
  double const radicand = d1 * d1 - dphi_0 * slope_a;
  if (radicand < 0.) return 0.5 * a;
  double const d2 = std::sqrt(radicand);
  double const denom = slope_a - dphi_0 + 2. * d2;
  if (denom == 0.) return 0.5 * a;
  return a - a * (slope_a + d2 - d1) / denom;
}
template <class Eval> static double line_search(LineSearchParams const& p, double phi_0, double dphi_0, Eval&& eval) {
  double const armijo_slope = p.c1 * dphi_0;
  double alpha = 1., best_alpha = 1., best_phi = std::numeric_limits<double>::max();
  for (int n = 1; n <= p.max_evals; ++n) {
    double phi, slope;
    if (!eval(alpha, phi, slope)) { alpha *= 0.5; continue; }
    if (phi < best_phi) { best_phi = phi; best_alpha = alpha; }
    if (phi <= phi_0 + alpha * armijo_slope) return alpha;
    double const alpha_model = ls_cubic_min(phi_0, dphi_0, alpha, phi, slope);
    alpha = std::min(std::max(alpha_model, p.backtrack_min * alpha), p.backtrack_max * alpha);
  }
  return best_alpha;
}

// ---------------------------------------------------------------------------
// Dense solve with complete pivoting and Eigen's rank rule
// (evaluations.cpp:112,456,624; small_J2.cpp:157 call Eigen fullPivLu().solve()).
// A is n x n row-major, B is n x m row-major; the solution overwrites X.
// A rank-deficient system returns the solution with free variables set to 0,
// which is what makes the `elastic` model's 1x1 zero dC/dxi yield dxi/dx = 0.
// ---------------------------------------------------------------------------
static void full_piv_lu_solve(int n, int m, double const* A_in, double const* B_in, double* X) {
  double A[8 * 8];
  std::vector<double> B(B_in, B_in + n * m);
  std::memcpy(A, A_in, sizeof(double) * n * n);
  int colperm[8];
  for (int i = 0; i < n; ++i) colperm[i] = i;
  double maxpivot = 0.;
  int rank = n;
  std::vector<double> piv(n, 0.);
  for (int k = 0; k < n; ++k) {
    int pr = k, pc = k;
    double big = 0.;
    for (int i = k; i < n; ++i)
      for (int j = k; j < n; ++j)
        if (std::fabs(A[i * n + j]) > big) { big = std::fabs(A[i * n + j]); pr = i; pc = j; }
    if (big == 0.) { rank = k; break; }
    if (big > maxpivot) maxpivot = big;
    if (pr != k) {
      for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[pr * n + j]);
      for (int j = 0; j < m; ++j) std::swap(B[k * m + j], B[pr * m + j]);
    }
    if (pc != k) {
      for (int i = 0; i < n; ++i) std::swap(A[i * n + k], A[i * n + pc]);
      std::swap(colperm[k], colperm[pc]);
    }
    piv[k] = A[k * n + k];
    for (int i = k + 1; i < n; ++i) {
      double const l = A[i * n + k] / A[k * n + k];
      A[i * n + k] = l;
      for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
      for (int j = 0; j < m; ++j) B[i * m + j] -= l * B[k * m + j];
    }
  }
  // Eigen: rank = #pivots with |pivot| > eps * n * maxpivot
  double const thresh = 2.220446049250313e-16 * n * maxpivot;
  int r = 0;
  for (int k = 0; k < rank; ++k) if (std::fabs(piv[k]) > thresh) ++r;
  rank = r;
  std::vector<double> Y(n * m, 0.);
  for (int k = rank - 1; k >= 0; --k) {
    for (int j = 0; j < m; ++j) {
      double s = B[k * m + j];
      for (int c = k + 1; c < rank; ++c) s -= A[k * n + c] * Y[c * m + j];
      Y[k * m + j] = s / A[k * n + k];
    }
  }
  for (int k = 0; k < n; ++k)
    for (int j = 0; j < m; ++j) X[colperm[k] * m + j] = Y[k * m + j];
}

// ---------------------------------------------------------------------------
// Element kit (weight.cpp:9-12, evaluations.cpp:82-85 -> apf getBF/getGradBF/
// getIntPoint/getIntWeight/getDV; mechanics.cpp:103-113 get_size).
// ---------------------------------------------------------------------------
enum { TRI3 = 3, TET4 = 4, HEX8 = 8 };

struct ElemKit {
  int type, nn, nedges;
  int edges[12][2];
  int npts[2];          // points in ip set 0 (coupled) and 1 (pressure)
  double pts[2][8][3];
  double wts[2][8];
};

static ElemKit make_kit(int type) {
  ElemKit k;
  std::memset(&k, 0, sizeof(k));
  k.type = type;
  if (type == TRI3) {  // the reference's 2-D element (disc.cpp:165)
    k.nn = 3;
    k.nedges = 3;
    int const e[3][2] = {{0, 1}, {1, 2}, {2, 0}};
    std::memcpy(k.edges, e, sizeof(e));
    // ip set 0: order 1: centroid, weight 1/2; ip set 1: order 2: three interior points, weight 1/6 each (apf's
    // triangle rules, third party; any degree-2 rule integrates the pressure-mass term exactly)
    k.npts[0] = 1;
    k.pts[0][0][0] = k.pts[0][0][1] = 1. / 3.;
    k.wts[0][0] = 0.5;
    k.npts[1] = 3;
    double const a = 1. / 6., b = 2. / 3.;
    double const q[3][2] = {{b, a}, {a, b}, {a, a}};
    for (int p = 0; p < 3; ++p) { k.pts[1][p][0] = q[p][0]; k.pts[1][p][1] = q[p][1]; k.wts[1][p] = 1. / 6.; }
  } else if (type == TET4) {
    k.nn = 4;
    k.nedges = 6;
    int const e[6][2] = {{0, 1}, {1, 2}, {2, 0}, {0, 3}, {1, 3}, {2, 3}};
    std::memcpy(k.edges, e, sizeof(e));
    // ip set 0: order 1 (mechanics.cpp:45): centroid, weight 1/6
    k.npts[0] = 1;
    k.pts[0][0][0] = k.pts[0][0][1] = k.pts[0][0][2] = 0.25;
    k.wts[0][0] = 1. / 6.;
    // ip set 1: order 2 (mechanics.cpp:46): 4-point rule
    double const a = 0.138196601125011, b = 0.585410196624969;
    k.npts[1] = 4;
    for (int p = 0; p < 4; ++p) {
      for (int d = 0; d < 3; ++d) k.pts[1][p][d] = a;
      if (p > 0) k.pts[1][p][p - 1] = b;
      k.wts[1][p] = 1. / 24.;
    }
  } else {
    k.nn = 8;
    k.nedges = 12;
    int const e[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                          {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    std::memcpy(k.edges, e, sizeof(e));
    // hex8 extension: both ip sets use the 2x2x2 Gauss-Legendre rule
    double const g = 0.5773502691896257645;
    for (int s = 0; s < 2; ++s) {
      k.npts[s] = 8;
      for (int p = 0; p < 8; ++p) {
        k.pts[s][p][0] = (p & 1) ? g : -g;
        k.pts[s][p][1] = (p & 2) ? g : -g;
        k.pts[s][p][2] = (p & 4) ? g : -g;
        k.wts[s][p] = 1.;
      }
    }
  }
  return k;
}

static int kit_dims(int type) { return type == TRI3 ? 2 : 3; }

static void shape(int type, double const* xi, double* N, double dNdxi[][3]) {
  if (type == TRI3) {
    N[0] = 1. - xi[0] - xi[1];
    N[1] = xi[0]; N[2] = xi[1];
    double const g[3][3] = {{-1, -1, 0}, {1, 0, 0}, {0, 1, 0}};
    std::memcpy(dNdxi, g, sizeof(g));
  } else if (type == TET4) {
    N[0] = 1. - xi[0] - xi[1] - xi[2];
    N[1] = xi[0]; N[2] = xi[1]; N[3] = xi[2];
    double const g[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    std::memcpy(dNdxi, g, sizeof(g));
  } else {
    static double const s[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1},
                                   {-1, -1, 1},  {1, -1, 1},  {1, 1, 1},  {-1, 1, 1}};
    for (int n = 0; n < 8; ++n) {
      double const a = 1. + s[n][0] * xi[0], b = 1. + s[n][1] * xi[1], c = 1. + s[n][2] * xi[2];
      N[n] = 0.125 * a * b * c;
      dNdxi[n][0] = 0.125 * s[n][0] * b * c;
      dNdxi[n][1] = 0.125 * s[n][1] * a * c;
      dNdxi[n][2] = 0.125 * s[n][2] * a * b;
    }
  }
}

// N, dN/dx and det(J) at a parametric point of an element with nodal coords X
static double shape_global(int type, int nn, double const X[][3], double const* xi,
                           double* N, double dN[][3]) {
  double dNdxi[8][3];
  shape(type, xi, N, dNdxi);
  int const nd = kit_dims(type);
  Tens<double> J;  // J(a,b) = d x_b / d xi_a
  J.dim = nd;
  for (int a = 0; a < nd; ++a) for (int b = 0; b < nd; ++b) {
    double s = 0.;
    for (int n = 0; n < nn; ++n) s += dNdxi[n][a] * X[n][b];
    J(a, b) = s;
  }
  Tens<double> const Ji = inverse(J);
  for (int n = 0; n < nn; ++n)
    for (int b = 0; b < 3; ++b) {
      double s = 0.;
      for (int a = 0; a < nd; ++a) s += Ji(b, a) * dNdxi[n][a];
      dN[n][b] = b < nd ? s : 0.;
    }
  return det(J);
}

// ---------------------------------------------------------------------------
// Global residual state + Mechanics (global_residual.cpp, mechanics.cpp).
// Two residuals: u (VECTOR, 3 eqs) and p (SCALAR, 1 eq); element DOF order is
// residual-major, node-minor: dx_idx = offset[i] + node*neq[i] + eq
// (global_residual.cpp:21-23).
// ---------------------------------------------------------------------------
enum { ELASTIC_PATH = 0, PLASTIC_PATH = 1 };

template <class T> struct Local;

template <class T> struct Global {
  int nn = 0, ndofs = 0, ndims = 3;
  int nres = 2;           // 2: mechanics (u, p); 1: mechanics_plane_stress (u only, mechanics_plane_stress.cpp:24-33)
  double thickness = 1.;  // mechanics_plane_stress.cpp:22
  int neq[2] = {3, 1};
  int off[2] = {0, 0};
  double stab_mult = 1.;
  double h = 0.;
  T x_nodal[2][8][3], x_prev_nodal[2][8][3], R_nodal[2][8][3];
  T x[2][3], x_prev[2][3], grad_x[2][3][3], grad_x_prev[2][3][3];
  double N[8], dN[8][3];
  Tens<T> F, F_prev, cof_F;
  T det_F;

  void before_elems(int nn_, int ndims_ = 3, int nres_ = 2) {  // global_residual.cpp:102-143; mechanics.cpp:16-55: u has ndims equations
    nn = nn_;
    ndims = ndims_;
    nres = nres_;
    neq[0] = ndims;
    off[0] = 0;
    off[1] = ndims * nn;
    ndofs = (ndims + (nres == 2 ? 1 : 0)) * nn;
    F.dim = F_prev.dim = cof_F.dim = ndims;
  }
  int dx_idx(int i, int n, int eq) const { return off[i] + n * neq[i] + eq; }

  void zero_residual() {  // :150-175
    for (int i = 0; i < nres; ++i) for (int n = 0; n < nn; ++n) for (int eq = 0; eq < neq[i]; ++eq)
      R_nodal[i][n][eq] = 0.;
  }
  void gather(double const* u, double const* p, double const* u_prev, double const* p_prev,
              int const* nodes) {  // :181-198 (assigning a double resets derivatives)
    for (int n = 0; n < nn; ++n) {
      for (int eq = 0; eq < ndims; ++eq) {
        x_nodal[0][n][eq] = u[nodes[n] * ndims + eq];
        x_prev_nodal[0][n][eq] = u_prev[nodes[n] * ndims + eq];
      }
      if (nres == 1) continue;
      x_nodal[1][n][0] = p[nodes[n]];
      x_prev_nodal[1][n][0] = p_prev[nodes[n]];
    }
  }
  int seed_wrt_x();       // :206-216
  void unseed_wrt_x();    // :227-239
  int seed_wrt_x_prev();  // :250-260
  void unseed_wrt_x_prev();

  void set_weights(double const* N_, double const dN_[][3]) {
    for (int n = 0; n < nn; ++n) { N[n] = N_[n]; for (int d = 0; d < 3; ++d) dN[n][d] = dN_[n][d]; }
  }
  void interpolate() {  // :289-332
    for (int i = 0; i < nres; ++i)
      for (int eq = 0; eq < neq[i]; ++eq) {
        x[i][eq] = x_nodal[i][0][eq] * N[0];
        x_prev[i][eq] = x_prev_nodal[i][0][eq] * N[0];
        for (int n = 1; n < nn; ++n) {
          x[i][eq] += x_nodal[i][n][eq] * N[n];
          x_prev[i][eq] += x_prev_nodal[i][n][eq] * N[n];
        }
      }
    for (int i = 0; i < nres; ++i)
      for (int eq = 0; eq < neq[i]; ++eq)
        for (int d = 0; d < ndims; ++d) {
          grad_x[i][eq][d] = x_nodal[i][0][eq] * dN[0][d];
          grad_x_prev[i][eq][d] = x_prev_nodal[i][0][eq] * dN[0][d];
          for (int n = 1; n < nn; ++n) {
            grad_x[i][eq][d] += x_nodal[i][n][eq] * dN[n][d];
            grad_x_prev[i][eq][d] += x_prev_nodal[i][n][eq] * dN[n][d];
          }
        }
    compute_kinematics();
  }
  void compute_kinematics() {  // mechanics.cpp:62-101
    for (int k = 0; k < ndims; ++k) {
      for (int l = 0; l < ndims; ++l) {
        F(k, l) = grad_x[0][k][l];
        F_prev(k, l) = grad_x_prev[0][k][l];
      }
      F(k, k) += T(1.0);
      F_prev(k, k) += T(1.0);
    }
    det_F = det(F);
    Tens<T>& C = cof_F;
    if (ndims == 2) {  // :95-100
      C(0, 0) = F(1, 1);
      C(0, 1) = -F(1, 0);
      C(1, 0) = -F(0, 1);
      C(1, 1) = F(0, 0);
      return;
    }
    C(0, 0) = F(1, 1) * F(2, 2) - F(1, 2) * F(2, 1);
    C(0, 1) = -F(1, 0) * F(2, 2) + F(1, 2) * F(2, 0);
    C(0, 2) = F(1, 0) * F(2, 1) - F(1, 1) * F(2, 0);
    C(1, 0) = -F(0, 1) * F(2, 2) + F(0, 2) * F(2, 1);
    C(1, 1) = F(0, 0) * F(2, 2) - F(0, 2) * F(2, 0);
    C(1, 2) = -F(0, 0) * F(2, 1) + F(0, 1) * F(2, 0);
    C(2, 0) = F(0, 1) * F(1, 2) - F(0, 2) * F(1, 1);
    C(2, 1) = -F(0, 0) * F(1, 2) + F(0, 2) * F(1, 0);
    C(2, 2) = F(0, 0) * F(1, 1) - F(0, 1) * F(1, 0);
  }
  T scalar_x(int i) const { return x[i][0]; }
  Vec<T> vector_x(int i) const { Vec<T> v; for (int d = 0; d < ndims; ++d) v(d) = x[i][d]; return v; }
  Vec<T> grad_scalar_x(int i) const { Vec<T> v; for (int d = 0; d < ndims; ++d) v(d) = grad_x[i][0][d]; return v; }
  Tens<T> grad_vector_x(int i) const {
    Tens<T> t; t.dim = ndims; for (int k = 0; k < ndims; ++k) for (int l = 0; l < ndims; ++l) t(k, l) = grad_x[i][k][l]; return t;
  }
  Tens<T> grad_vector_x_prev(int i) const {
    Tens<T> t; t.dim = ndims; for (int k = 0; k < ndims; ++k) for (int l = 0; l < ndims; ++l) t(k, l) = grad_x_prev[i][k][l]; return t;
  }

  // mechanics.cpp:116-145, 148-227, 230-240 (mixed formulation)
  void evaluate(Local<T>& local, double w, double dv, int ip_set);

  void residual_values(double* R) const {  // :380-391
    for (int i = 0; i < nres; ++i) for (int n = 0; n < nn; ++n) for (int eq = 0; eq < neq[i]; ++eq)
      R[dx_idx(i, n, eq)] = val(R_nodal[i][n][eq]);
  }
  void jacobian(int nderivs, double* J) const;  // :400-414, row-major ndofs x nderivs
};

template <> int Global<double>::seed_wrt_x() { return -1; }
template <> void Global<double>::unseed_wrt_x() {}
template <> int Global<double>::seed_wrt_x_prev() { return -1; }
template <> void Global<double>::unseed_wrt_x_prev() {}
template <> void Global<double>::jacobian(int, double*) const {}
template <> int Global<Fad>::seed_wrt_x() {
  for (int i = 0; i < nres; ++i) for (int n = 0; n < nn; ++n) for (int eq = 0; eq < neq[i]; ++eq)
    x_nodal[i][n][eq].diff(dx_idx(i, n, eq), ndofs);
  return ndofs;
}
template <> void Global<Fad>::unseed_wrt_x() {
  for (int i = 0; i < nres; ++i) for (int n = 0; n < nn; ++n) for (int eq = 0; eq < neq[i]; ++eq) {
    x_nodal[i][n][eq] = x_nodal[i][n][eq].val();
    R_nodal[i][n][eq].n = 0;  // derivative slots of R are zeroed too (:234)
  }
}
template <> int Global<Fad>::seed_wrt_x_prev() {
  for (int i = 0; i < nres; ++i) for (int n = 0; n < nn; ++n) for (int eq = 0; eq < neq[i]; ++eq)
    x_prev_nodal[i][n][eq].diff(dx_idx(i, n, eq), ndofs);
  return ndofs;
}
template <> void Global<Fad>::unseed_wrt_x_prev() {
  for (int i = 0; i < nres; ++i) for (int n = 0; n < nn; ++n) for (int eq = 0; eq < neq[i]; ++eq) {
    x_prev_nodal[i][n][eq] = x_prev_nodal[i][n][eq].val();
    R_nodal[i][n][eq].n = 0;
  }
}
template <> void Global<Fad>::jacobian(int nderivs, double* J) const {
  for (int i = 0; i < nres; ++i) for (int n = 0; n < nn; ++n) for (int eq = 0; eq < neq[i]; ++eq) {
    int const r = dx_idx(i, n, eq);
    for (int j = 0; j < nderivs; ++j) J[r * nderivs + j] = R_nodal[i][n][eq].dx(j);
  }
}

// ---------------------------------------------------------------------------
// Local residual base (local_residual.cpp) and the three hot models.
// ---------------------------------------------------------------------------
template <class T> T compute_mu(T const& E, T const& nu) { return E / (2. * (1. + nu)); }      // material_params.hpp:12
template <class T> T compute_kappa(T const& E, T const& nu) { return E / (3. * (1. - 2. * (nu))); }  // :20

template <class T> struct Local {
  int nres = 0, ndims = 3;
  int neq[4] = {0, 0, 0, 0};
  int off[4] = {0, 0, 0, 0};
  int z_stretch_idx = -1;  // local_residual.hpp:423,452: the out-of-plane stretch of the finite-deformation plane-stress models
  int ndofs = 0;
  int max_iters = 0;
  double abs_tol = 0., rel_tol = 0.;
  std::vector<T> params;
  T xi[8], xi_prev[8], R[8];
  virtual ~Local() {}

  void finish_layout() {  // local_residual.cpp:88-94
    ndofs = 0;
    for (int i = 0; i < nres; ++i) { off[i] = ndofs; ndofs += neq[i]; }
  }
  void before_elems(double const* p, int np) {  // :96-100
    params.resize(np);
    for (int k = 0; k < np; ++k) params[k] = p[k];
  }
  double norm_residual() const {  // :110-120
    double nrm = 0.;
    for (int k = 0; k < ndofs; ++k) { double const v = val(R[k]); nrm += v * v; }
    return std::sqrt(nrm);
  }
  void gather(double const* xi_pt, double const* xi_prev_pt) {  // :599-617
    for (int k = 0; k < ndofs; ++k) { R[k] = 0.; xi[k] = xi_pt[k]; xi_prev[k] = xi_prev_pt[k]; }
  }
  void scatter(double* xi_pt) const { for (int k = 0; k < ndofs; ++k) xi_pt[k] = val(xi[k]); }  // :624-631
  int seed_wrt_xi();            // :703-711
  void unseed_wrt_xi();         // :723-733
  int seed_wrt_xi_prev();       // :745-753
  void unseed_wrt_xi_prev();    // :765-775
  void seed_wrt_x(int nglobal, double const* dxi_dx);  // :786-800
  int seed_wrt_params(int nactive, int const* active);    // :812-819
  void unseed_wrt_params(int nactive, int const* active); // :831-845
  void jacobian(int nderivs, double* J) const;  // :129-141
  void residual_values(double* r) const { for (int k = 0; k < ndofs; ++k) r[k] = val(R[k]); }  // :164-173

  // symmetric tensors are packed (00,01,02,11,12,22) (:206-216, :336-341, :572-577)
  Tens<T> sym(T const* s) const {
    Tens<T> t;
    t.dim = ndims;
    if (ndims == 2) {  // (00,01,11)
      t(0, 0) = s[0]; t(0, 1) = s[1];
      t(1, 0) = s[1]; t(1, 1) = s[2];
      return t;
    }
    t(0, 0) = s[0]; t(0, 1) = s[1]; t(0, 2) = s[2];
    t(1, 0) = s[1]; t(1, 1) = s[3]; t(1, 2) = s[4];
    t(2, 0) = s[2]; t(2, 1) = s[4]; t(2, 2) = s[5];
    return t;
  }
  Tens<T> sym_tensor_xi(int i) const { return sym(&xi[off[i]]); }
  Tens<T> sym_tensor_xi_prev(int i) const { return sym(&xi_prev[off[i]]); }
  T scalar_xi(int i) const { return xi[off[i]]; }
  T scalar_xi_prev(int i) const { return xi_prev[off[i]]; }
  void set_scalar_xi_val(int i, double v);            // value only, keeps seeding (:293-296)
  void set_sym_tensor_xi_val(int i, Tens<T> const& t);  // (:346-360)
  void add_to_xi(double const* dxi);                  // (:420-424, :478-492)
  void set_sym_tensor_R(int i, Tens<T> const& t) {  // :565-579
    T* r = &R[off[i]];
    if (ndims == 2) { r[0] = t(0, 0); r[1] = t(0, 1); r[2] = t(1, 1); return; }
    r[0] = t(0, 0); r[1] = t(0, 1); r[2] = t(0, 2); r[3] = t(1, 1); r[4] = t(1, 2); r[5] = t(2, 2);
  }
  void set_scalar_R(int i, T const& v) { R[off[i]] = v; }

  virtual int num_params() const = 0;
  virtual void init_variables(double* xi_pt) const = 0;
  virtual bool is_finite_deformation() const = 0;
  virtual int solve_nonlinear(Global<T>& g) = 0;
  virtual int evaluate(Global<T>& g, bool force_path = false, int path_in = 0) = 0;
  virtual Tens<T> cauchy(Global<T>& g) = 0;
  virtual Tens<T> dev_cauchy(Global<T>& g) = 0;
  virtual T hydro_cauchy(Global<T>& g) = 0;
  virtual T pressure_scale_factor() = 0;

  // The Newton iteration shared by small_J2.cpp:137-171 and hyper_J2.cpp:181-216
  int newton(Global<T>& g);
  // The Newton iteration of the Hosford / Barlat models (small_hosford.cpp:147-218, hypo_hosford.cpp:183-254,
  // hypo_barlat.cpp:353-432): the branch is chosen by the first evaluation and forced afterwards, and every step is
  // followed by the line search of line_search.hpp on the merit 1/2 |C|^2
  LineSearchParams ls;
  int newton_ls(Global<T>& g);
};

template <> int Local<double>::seed_wrt_xi() { return -1; }
template <> void Local<double>::unseed_wrt_xi() {}
template <> int Local<double>::seed_wrt_xi_prev() { return -1; }
template <> void Local<double>::unseed_wrt_xi_prev() {}
template <> void Local<double>::seed_wrt_x(int, double const*) {}
template <> int Local<double>::seed_wrt_params(int, int const*) { return -1; }
template <> void Local<double>::unseed_wrt_params(int, int const*) {}
template <> void Local<double>::jacobian(int, double*) const {}
template <> void Local<double>::set_scalar_xi_val(int i, double v) { xi[off[i]] = v; }
template <> void Local<double>::set_sym_tensor_xi_val(int i, Tens<double> const& t) {
  double* s = &xi[off[i]];
  if (ndims == 2) { s[0] = t(0, 0); s[1] = t(0, 1); s[2] = t(1, 1); return; }
  s[0] = t(0, 0); s[1] = t(0, 1); s[2] = t(0, 2); s[3] = t(1, 1); s[4] = t(1, 2); s[5] = t(2, 2);
}
template <> void Local<double>::add_to_xi(double const* dxi) { for (int k = 0; k < ndofs; ++k) xi[k] += dxi[k]; }
template <> int Local<double>::newton(Global<double>&) { return 0; }
template <> int Local<double>::newton_ls(Global<double>&) { return 0; }

template <> int Local<Fad>::seed_wrt_xi() {
  for (int k = 0; k < ndofs; ++k) xi[k].diff(k, ndofs);
  return ndofs;
}
template <> void Local<Fad>::unseed_wrt_xi() {
  for (int k = 0; k < ndofs; ++k) { xi[k] = xi[k].val(); R[k].n = 0; }
}
template <> int Local<Fad>::seed_wrt_xi_prev() {
  for (int k = 0; k < ndofs; ++k) xi_prev[k].diff(k, ndofs);
  return ndofs;
}
template <> void Local<Fad>::unseed_wrt_xi_prev() {
  for (int k = 0; k < ndofs; ++k) { xi_prev[k] = xi_prev[k].val(); R[k].n = 0; }
}
template <> void Local<Fad>::seed_wrt_x(int nglobal, double const* dxi_dx) {
  for (int k = 0; k < ndofs; ++k) {
    double const v = xi[k].val();
    xi[k].diff(0, nglobal);
    xi[k].v = v;
    for (int j = 0; j < nglobal; ++j) xi[k].d[j] = dxi_dx[k * nglobal + j];
  }
}
template <> int Local<Fad>::seed_wrt_params(int nactive, int const* active) {
  for (int p = 0; p < nactive; ++p) params[active[p]].diff(p, nactive);
  return nactive;
}
template <> void Local<Fad>::unseed_wrt_params(int nactive, int const* active) {
  for (int p = 0; p < nactive; ++p) params[active[p]] = params[active[p]].val();
  for (int k = 0; k < ndofs; ++k) R[k].n = 0;
}
template <> void Local<Fad>::jacobian(int nderivs, double* J) const {
  for (int k = 0; k < ndofs; ++k) for (int j = 0; j < nderivs; ++j) J[k * nderivs + j] = R[k].dx(j);
}
template <> void Local<Fad>::set_scalar_xi_val(int i, double v) { xi[off[i]].v = v; }
template <> void Local<Fad>::set_sym_tensor_xi_val(int i, Tens<Fad> const& t) {
  Fad* s = &xi[off[i]];
  if (ndims == 2) { s[0].v = t(0, 0).v; s[1].v = t(0, 1).v; s[2].v = t(1, 1).v; return; }
  s[0].v = t(0, 0).v; s[1].v = t(0, 1).v; s[2].v = t(0, 2).v;
  s[3].v = t(1, 1).v; s[4].v = t(1, 2).v; s[5].v = t(2, 2).v;
}
template <> void Local<Fad>::add_to_xi(double const* dxi) { for (int k = 0; k < ndofs; ++k) xi[k].v += dxi[k]; }

template <> int Local<Fad>::newton(Global<Fad>& g) {
  int path = ELASTIC_PATH;
  int iter = 1;
  double R_norm_0 = 1.;
  bool converged = false;
  while ((iter <= max_iters) && (!converged)) {
    path = this->evaluate(g);
    double const R_norm = this->norm_residual();
    if (iter == 1) R_norm_0 = R_norm;
    double const R_norm_rel = R_norm / R_norm_0;  // 0/0 = NaN on elastic points: the abs test decides
    if ((R_norm_rel < rel_tol) || (R_norm < abs_tol)) { converged = true; break; }
    double J[64], r[8], dxi[8];
    this->jacobian(ndofs, J);
    this->residual_values(r);
    for (int k = 0; k < ndofs; ++k) r[k] = -r[k];
    full_piv_lu_solve(ndofs, 1, J, r, dxi);
    this->add_to_xi(dxi);
    iter++;
  }
  if ((iter > max_iters) && (!converged)) return -1;
  return path;
}

template <> int Local<Fad>::newton_ls(Global<Fad>& g) {
  int path = ELASTIC_PATH;
  int iter = 1;
  double C_norm_0 = 1.;
  bool converged = false;
  while ((iter <= max_iters) && (!converged)) {
    if (iter == 1) path = this->evaluate(g);
    else this->evaluate(g, true, path);
    double const C_norm = this->norm_residual();
    if (iter == 1) C_norm_0 = C_norm;
    double const C_norm_rel = C_norm / C_norm_0;
    if ((C_norm_rel < rel_tol) || (C_norm < abs_tol)) { converged = true; break; }
    double J[64], r[8], dxi[8], step[8];
    this->jacobian(ndofs, J);
    this->residual_values(r);
    for (int k = 0; k < ndofs; ++k) r[k] = -r[k];
    full_piv_lu_solve(ndofs, 1, J, r, dxi);
    this->add_to_xi(dxi);
    {
      double const psi_0 = 0.5 * C_norm * C_norm;
      double const dpsi_0 = -2. * psi_0;
      double alpha_applied = 1.;  // the full Newton step was applied above
      auto move = [&](double alpha) {
        double const alpha_diff = alpha - alpha_applied;
        alpha_applied = alpha;
        for (int k = 0; k < ndofs; ++k) step[k] = alpha_diff * dxi[k];
        this->add_to_xi(step);
      };
      auto eval = [&](double alpha, double& phi, double& slope) -> bool {
        move(alpha);
        path = this->evaluate(g, true, path);
        double const C_alpha = this->norm_residual();
        phi = 0.5 * C_alpha * C_alpha;
        double Ja[64], Ca[8];
        this->jacobian(ndofs, Ja);
        this->residual_values(Ca);
        slope = 0.;  // phi'(alpha) = C . (J dxi)
        for (int i = 0; i < ndofs; ++i) {
          double Jd = 0.;
          for (int k = 0; k < ndofs; ++k) Jd += Ja[i * ndofs + k] * dxi[k];
          slope += Ca[i] * Jd;
        }
        return true;
      };
      move(line_search(ls, psi_0, dpsi_0, eval));
    }
    iter++;
  }
  if ((iter > max_iters) && (!converged)) return -1;
  return path;
}

// elastic.cpp:76-136 (one dummy scalar local variable; params E, nu, cte, delta_T)
template <class T> struct Elastic : Local<T> {
  Elastic() { this->nres = 1; this->neq[0] = 1; this->finish_layout(); }
  int num_params() const override { return 4; }
  void init_variables(double* xi_pt) const override { xi_pt[0] = 0.; }
  bool is_finite_deformation() const override { return false; }
  int solve_nonlinear(Global<T>&) override { this->set_scalar_xi_val(0, 0.); return 0; }
  int evaluate(Global<T>&, bool, int) override { return 0; }
  Tens<T> cauchy(Global<T>& g) override {
    T const p = g.scalar_x(1);
    Tens<T> const I = eye<T>();
    Tens<T> const dev_sigma = this->dev_cauchy(g);
    return dev_sigma - p * I;
  }
  Tens<T> dev_cauchy(Global<T>& g) override {
    Tens<T> const I = eye<T>();
    T const E = this->params[0];
    T const nu = this->params[1];
    T const mu = compute_mu(E, nu);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    Tens<T> const dev_eps = eps - (trace(eps) / 3.) * I;
    return (2. * mu) * dev_eps;
  }
  T hydro_cauchy(Global<T>& g) override {
    T const E = this->params[0];
    T const nu = this->params[1];
    T const kappa = compute_kappa(E, nu);
    T const cte = this->params[2];
    T const delta_T = this->params[3];
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    return kappa * trace(eps) - cte * delta_T * E / (1. - 2. * nu);
  }
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

// small_J2.cpp (pstrain SYM_TENSOR + alpha SCALAR; params E, nu, K, Y, cte, delta_T)
template <class T> struct SmallJ2 : Local<T> {
  explicit SmallJ2(int ndims = 3) {  // get_num_eqs(SYM_TENSOR, ndims) = 6 or 3 (small_J2.cpp:45-51)
    this->ndims = ndims;
    this->nres = 2; this->neq[0] = (ndims == 3) ? 6 : 3; this->neq[1] = 1; this->finish_layout();
  }
  int num_params() const override { return 6; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < this->ndofs; ++k) xi_pt[k] = 0.; }
  bool is_finite_deformation() const override { return false; }
  int solve_nonlinear(Global<T>& g) override {  // :122-173
    if (std::is_same<T, double>::value) return 0;
    {
      Tens<T> const pstrain_old = this->sym_tensor_xi_prev(0);
      T const alpha_old = this->scalar_xi_prev(1);
      this->set_sym_tensor_xi_val(0, pstrain_old);
      this->set_scalar_xi_val(1, val(alpha_old));
    }
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :181-250
    int path = ELASTIC_PATH;
    double const sqrt_23 = std::sqrt(2. / 3.);
    double const sqrt_32 = std::sqrt(3. / 2.);
    T const E = this->params[0];
    T const nu = this->params[1];
    T const K = this->params[2];
    T const Y = this->params[3];
    T const mu = compute_mu(E, nu);
    Tens<T> const pstrain_old = this->sym_tensor_xi_prev(0);
    T const alpha_old = this->scalar_xi_prev(1);
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    T const alpha = this->scalar_xi(1);
    Tens<T> const s = this->dev_cauchy(g);
    T const s_mag = norm(s);
    Tens<T> const n = s / s_mag;  // 0/0 at zero strain; only read on the plastic branch
    T const sigma_yield = Y + K * alpha;
    T const f = (s_mag - sqrt_23 * sigma_yield) / val(mu);
    Tens<T> R_pstrain;
    T R_alpha;
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    if (plastic) {
      T const dgam = sqrt_32 * (alpha - alpha_old);
      R_pstrain = pstrain - pstrain_old - dgam * n;
      R_alpha = f;
    } else {
      R_pstrain = pstrain - pstrain_old;
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_pstrain);
    this->set_scalar_R(1, R_alpha);
    return path;
  }
  Tens<T> cauchy(Global<T>& g) override {  // :253-263
    T const p = g.scalar_x(1);
    Tens<T> const I = eye<T>(g.ndims);
    Tens<T> const dev_sigma = this->dev_cauchy(g);
    return dev_sigma - p * I;
  }
  Tens<T> dev_cauchy(Global<T>& g) override {  // :266-277
    Tens<T> const I = eye<T>(g.ndims);
    T const E = this->params[0];
    T const nu = this->params[1];
    T const mu = compute_mu(E, nu);
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    Tens<T> const dev_eps = eps - (trace(eps) / 3.) * I;
    return (2. * mu) * (dev_eps - pstrain);
  }
  T hydro_cauchy(Global<T>& g) override {  // :280-289
    T const E = this->params[0];
    T const nu = this->params[1];
    T const kappa = compute_kappa(E, nu);
    T const cte = this->params[4];
    T const delta_T = this->params[5];
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    return kappa * trace(eps) - cte * delta_T * E / (1. - 2. * nu);
  }
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

// isotropic_elastic.cpp, mixed formulation (cauchy SYM_TENSOR; params E nu)
template <class T> static T compute_lambda(T const& E, T const& nu) { return E * nu / ((1. + nu) * (1. - 2. * nu)); }  // material_params.hpp:28
template <class T> struct IsotropicElastic : Local<T> {
  IsotropicElastic() { this->nres = 1; this->neq[0] = 6; this->finish_layout(); }
  int num_params() const override { return 2; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 6; ++k) xi_pt[k] = 0.; }
  bool is_finite_deformation() const override { return false; }
  Tens<T> hooke(Global<T>& g) {
    T const mu = compute_mu(this->params[0], this->params[1]);
    T const lambda = compute_lambda(this->params[0], this->params[1]);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    return (lambda * trace(eps)) * eye<T>() + (2. * mu) * eps;
  }
  int solve_nonlinear(Global<T>& g) override {  // :93-122: exact initial guess, then exactly one Newton step
    if (std::is_same<T, double>::value) return 0;
    this->set_sym_tensor_xi_val(0, hooke(g));
    int const path = this->evaluate(g, false, 0);
    double J[64], r[8], dxi[8];
    this->jacobian(this->ndofs, J);
    this->residual_values(r);
    for (int k = 0; k < this->ndofs; ++k) r[k] = -r[k];
    full_piv_lu_solve(this->ndofs, 1, J, r, dxi);
    this->add_to_xi(dxi);
    return path;
  }
  int evaluate(Global<T>& g, bool, int) override {  // :128-152
    Tens<T> const R_cauchy = this->sym_tensor_xi(0) - hooke(g);
    this->set_sym_tensor_R(0, R_cauchy);
    return 0;
  }
  T hydro_cauchy(Global<T>&) override { return trace(this->sym_tensor_xi(0)) / 3.; }  // :170-181
  Tens<T> dev_cauchy(Global<T>& g) override { return this->sym_tensor_xi(0) - this->hydro_cauchy(g) * eye<T>(); }  // :162-168
  Tens<T> cauchy(Global<T>& g) override { return this->dev_cauchy(g) - g.scalar_x(1) * eye<T>(); }  // :191-197
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

// small_hill.cpp (pstrain SYM_TENSOR, alpha SCALAR; params E nu Y R00 R11 R22 R01 R02 R12 S D);
// Hill's yield function and its normal: yield_functions.hpp:34-99
template <class T> struct SmallHill : Local<T> {
  SmallHill() { this->nres = 2; this->neq[0] = 6; this->neq[1] = 1; this->finish_layout(); }
  int num_params() const override { return 11; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 7; ++k) xi_pt[k] = 0.; }
  bool is_finite_deformation() const override { return false; }
  int solve_nonlinear(Global<T>& g) override {  // :137-190
    if (std::is_same<T, double>::value) return 0;
    {
      Tens<T> const pstrain_old = this->sym_tensor_xi_prev(0);
      T const alpha_old = this->scalar_xi_prev(1);
      this->set_sym_tensor_xi_val(0, pstrain_old);
      this->set_scalar_xi_val(1, val(alpha_old));
    }
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :196-268
    int path = ELASTIC_PATH;
    T const E = this->params[0], nu = this->params[1], Y = this->params[2];
    T const R00 = this->params[3], R11 = this->params[4], R22 = this->params[5];
    T const R01 = this->params[6], R02 = this->params[7], R12 = this->params[8];
    T const S = this->params[9], D = this->params[10];
    T const mu = compute_mu(E, nu);
    auto inv2 = [](T const& r) { return 1. / (r * r); };  // std::pow(r, -2)
    T hp[6];  // compute_hill_params
    hp[0] = 0.5 * (inv2(R11) + inv2(R22) - inv2(R00));
    hp[1] = 0.5 * (inv2(R22) + inv2(R00) - inv2(R11));
    hp[2] = 0.5 * (inv2(R00) + inv2(R11) - inv2(R22));
    hp[3] = 1.5 * inv2(R12);
    hp[4] = 1.5 * inv2(R02);
    hp[5] = 1.5 * inv2(R01);
    Tens<T> const pstrain_old = this->sym_tensor_xi_prev(0);
    T const alpha_old = this->scalar_xi_prev(1);
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    T const alpha = this->scalar_xi(1);
    Tens<T> const s = this->dev_cauchy(g);
    T const d12 = s(1, 1) - s(2, 2), d20 = s(2, 2) - s(0, 0), d01 = s(0, 0) - s(1, 1);
    T const hill = sqrt(hp[0] * d12 * d12 + hp[1] * d20 * d20 + hp[2] * d01 * d01 +
                        2. * (hp[3] * s(1, 2) * s(1, 2) + hp[4] * s(0, 2) * s(0, 2) + hp[5] * s(0, 1) * s(0, 1)));
    T const sigma_yield = Y + S * (1. - exp(-(D * alpha)));
    T const f = (hill - sigma_yield) / val(mu);
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    Tens<T> R_pstrain;
    T R_alpha;
    if (plastic) {
      Tens<T> n;  // compute_hill_normal
      n(0, 0) = (hp[1] + hp[2]) * s(0, 0) - hp[2] * s(1, 1) - hp[1] * s(2, 2);
      n(1, 1) = (hp[0] + hp[2]) * s(1, 1) - hp[2] * s(0, 0) - hp[0] * s(2, 2);
      n(2, 2) = (hp[1] + hp[0]) * s(2, 2) - hp[1] * s(0, 0) - hp[0] * s(1, 1);
      n(0, 1) = hp[5] * s(0, 1); n(0, 2) = hp[4] * s(0, 2); n(1, 2) = hp[3] * s(1, 2);
      n(1, 0) = n(0, 1); n(2, 0) = n(0, 2); n(2, 1) = n(1, 2);
      n = n / hill;
      T const dgam = alpha - alpha_old;
      R_pstrain = pstrain - pstrain_old - dgam * n;
      R_pstrain(2, 2) = trace(pstrain);
      R_alpha = f;
    } else {
      R_pstrain = pstrain - pstrain_old;
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_pstrain);
    this->set_scalar_R(1, R_alpha);
    return path;
  }
  Tens<T> cauchy(Global<T>& g) override {  // :271-281
    T const p = g.scalar_x(1);
    return this->dev_cauchy(g) - p * eye<T>();
  }
  Tens<T> dev_cauchy(Global<T>& g) override {  // :284-295
    Tens<T> const I = eye<T>();
    T const mu = compute_mu(this->params[0], this->params[1]);
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    Tens<T> const dev_eps = eps - (trace(eps) / 3.) * I;
    return (2. * mu) * (dev_eps - pstrain);
  }
  T hydro_cauchy(Global<T>& g) override {  // :298-306
    T const kappa = compute_kappa(this->params[0], this->params[1]);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    return kappa * trace(eps);
  }
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

// small_hill_plane_strain.cpp (2-D: pstrain SYM_TENSOR (00,01,11), alpha SCALAR; params E nu Y S D R00 R11 R22 R01;
// R02 = R12 = 1): the in-plane deviatoric stress is completed with s_zz = 2 mu (-tr(eps)/3 + tr(pstrain)) for the Hill
// function (:226-233); the flow direction is the in-plane part of the 3-D Hill normal (:243-247); no equation is replaced
template <class T> struct SmallHillPlaneStrain : Local<T> {
  SmallHillPlaneStrain() { this->ndims = 2; this->nres = 2; this->neq[0] = 3; this->neq[1] = 1; this->finish_layout(); }
  int num_params() const override { return 9; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 4; ++k) xi_pt[k] = 0.; }
  bool is_finite_deformation() const override { return false; }
  int solve_nonlinear(Global<T>& g) override {  // :133-185
    if (std::is_same<T, double>::value) return 0;
    {
      Tens<T> const pstrain_old = this->sym_tensor_xi_prev(0);
      T const alpha_old = this->scalar_xi_prev(1);
      this->set_sym_tensor_xi_val(0, pstrain_old);
      this->set_scalar_xi_val(1, val(alpha_old));
    }
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :193-277
    int path = ELASTIC_PATH;
    T const E = this->params[0], nu = this->params[1], Y = this->params[2], S = this->params[3], D = this->params[4];
    T const R00 = this->params[5], R11 = this->params[6], R22 = this->params[7], R01 = this->params[8];
    T const R02 = 1., R12 = 1.;
    T const mu = compute_mu(E, nu);
    auto inv2 = [](T const& r) { return 1. / (r * r); };  // std::pow(r, -2)
    T hp[6];  // compute_hill_params (yield_functions.hpp:35-50)
    hp[0] = 0.5 * (inv2(R11) + inv2(R22) - inv2(R00));
    hp[1] = 0.5 * (inv2(R22) + inv2(R00) - inv2(R11));
    hp[2] = 0.5 * (inv2(R00) + inv2(R11) - inv2(R22));
    hp[3] = 1.5 * inv2(R12);
    hp[4] = 1.5 * inv2(R02);
    hp[5] = 1.5 * inv2(R01);
    Tens<T> const pstrain_old = this->sym_tensor_xi_prev(0);
    T const alpha_old = this->scalar_xi_prev(1);
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    T const alpha = this->scalar_xi(1);
    Tens<T> const s_2D = this->dev_cauchy(g);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const epsilon = 0.5 * (grad_u + transpose(grad_u));
    T const s_zz = 2. * mu * (-trace(epsilon) / 3. + trace(pstrain));
    Tens<T> s = s_2D;  // insert_2D_tensor_into_3D
    s.dim = 3;
    s(2, 2) = s_zz;
    T const d12 = s(1, 1) - s(2, 2), d20 = s(2, 2) - s(0, 0), d01 = s(0, 0) - s(1, 1);
    T const hill = sqrt(hp[0] * d12 * d12 + hp[1] * d20 * d20 + hp[2] * d01 * d01 +
                        2. * (hp[3] * s(1, 2) * s(1, 2) + hp[4] * s(0, 2) * s(0, 2) + hp[5] * s(0, 1) * s(0, 1)));
    T const sigma_yield = Y + S * (1. - exp(-(D * alpha)));
    T const f = (hill - sigma_yield) / val(mu);
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    Tens<T> R_pstrain;
    T R_alpha;
    if (plastic) {
      Tens<T> n;  // in-plane part of compute_hill_normal
      n.dim = 2;
      n(0, 0) = ((hp[1] + hp[2]) * s(0, 0) - hp[2] * s(1, 1) - hp[1] * s(2, 2)) / hill;
      n(1, 1) = ((hp[0] + hp[2]) * s(1, 1) - hp[2] * s(0, 0) - hp[0] * s(2, 2)) / hill;
      n(0, 1) = (hp[5] * s(0, 1)) / hill;
      n(1, 0) = n(0, 1);
      T const dgam = alpha - alpha_old;
      R_pstrain = pstrain - pstrain_old - dgam * n;
      R_alpha = f;
    } else {
      R_pstrain = pstrain - pstrain_old;
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_pstrain);
    this->set_scalar_R(1, R_alpha);
    return path;
  }
  Tens<T> cauchy(Global<T>& g) override {  // :280-290
    T const p = g.scalar_x(1);
    return this->dev_cauchy(g) - p * eye<T>(g.ndims);
  }
  Tens<T> dev_cauchy(Global<T>& g) override {  // :293-304
    Tens<T> const I = eye<T>(g.ndims);
    T const mu = compute_mu(this->params[0], this->params[1]);
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    Tens<T> const dev_eps = eps - (trace(eps) / 3.) * I;
    return (2. * mu) * (dev_eps - pstrain);
  }
  T hydro_cauchy(Global<T>& g) override {  // :307-315
    T const kappa = compute_kappa(this->params[0], this->params[1]);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    return kappa * trace(eps);
  }
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

// hyper_J2_plane_strain.cpp (2-D: zeta SYM_TENSOR (00,01,11), Ie SCALAR, alpha SCALAR; params E nu K Y Y_inf delta):
// finite-deformation J2 in plane strain.  The in-plane tensors are completed out of plane where the 3-D quantity is
// needed: zeta_zz = -tr(zeta) (:153, :256), be_bar_zz = (zeta_zz + Ie) / det(rF)^(2/3) (:154).
template <class T> struct HyperJ2PlaneStrain : Local<T> {
  HyperJ2PlaneStrain() { this->ndims = 2; this->nres = 3; this->neq[0] = 3; this->neq[1] = 1; this->neq[2] = 1; this->finish_layout(); }
  int num_params() const override { return 6; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 5; ++k) xi_pt[k] = 0.; xi_pt[3] = 1.; }  // :119-131
  bool is_finite_deformation() const override { return true; }
  // eval_be_bar_plane_strain (:134-156): the 3 x 3 trial tensor, in-plane block + (2,2) entry
  Tens<T> be_bar(Global<T>& g, Tens<T> const& zeta, T const& Ie) const {
    Tens<T> const I = eye<T>(2);
    Tens<T> const F = g.grad_vector_x(0) + I;
    Tens<T> const F_prev = g.grad_vector_x_prev(0) + I;
    Tens<T> const rF = F * inverse(F_prev);
    T const det_rF = det(rF);
    T const det_rF_13 = cbrt(det_rF);
    Tens<T> const rF_bar = rF / det_rF_13;
    Tens<T> const rF_barT = transpose(rF_bar);
    Tens<T> const be_bar_2D = rF_bar * (zeta + Ie * I) * rF_barT;
    T const zeta_zz = -trace(zeta);
    T const be_bar_zz = (zeta_zz + Ie) / (det_rF_13 * det_rF_13);
    Tens<T> be = be_bar_2D;  // insert_2D_tensor_into_3D
    be.dim = 3;
    be(2, 2) = be_bar_zz;
    return be;
  }
  static Tens<T> in_plane(Tens<T> const& t3) {  // extract_2D_tensor_from_3D
    Tens<T> t;
    t.dim = 2;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) t(i, j) = t3(i, j);
    return t;
  }
  int solve_nonlinear(Global<T>& g) override {  // :163-223
    if (std::is_same<T, double>::value) return 0;
    {
      Tens<T> const zeta_old = this->sym_tensor_xi_prev(0);
      T const Ie_old = this->scalar_xi_prev(1);
      T const alpha_old = this->scalar_xi_prev(2);
      Tens<T> const bt = be_bar(g, zeta_old, Ie_old);
      T const Ie_trial = trace(bt) / 3.;
      Tens<T> const zeta_trial = in_plane(bt) - Ie_trial * eye<T>(2);
      this->set_sym_tensor_xi_val(0, zeta_trial);
      this->set_scalar_xi_val(1, val(Ie_trial));
      this->set_scalar_xi_val(2, val(alpha_old));
    }
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :231-325
    int path = ELASTIC_PATH;
    double const sqrt_23 = std::sqrt(2. / 3.);
    double const sqrt_32 = std::sqrt(3. / 2.);
    T const E = this->params[0], nu = this->params[1], K = this->params[2], Y = this->params[3];
    T const Y_inf = this->params[4], delta = this->params[5];
    T const mu = compute_mu(E, nu);
    Tens<T> const zeta_old = this->sym_tensor_xi_prev(0);
    T const Ie_old = this->scalar_xi_prev(1);
    T const alpha_old = this->scalar_xi_prev(2);
    Tens<T> const zeta = this->sym_tensor_xi(0);
    T const Ie = this->scalar_xi(1);
    T const alpha = this->scalar_xi(2);
    Tens<T> const I = eye<T>(2);
    Tens<T> const bt = be_bar(g, zeta_old, Ie_old);
    T const Ie_trial = trace(bt) / 3.;
    Tens<T> const zeta_trial = in_plane(bt) - Ie_trial * I;
    Tens<T> zeta_3D = zeta;
    zeta_3D.dim = 3;
    zeta_3D(2, 2) = -trace(zeta);
    Tens<T> const be_bar_3D = zeta_3D + Ie * eye<T>(3);
    Tens<T> const s_3D = mu * zeta_3D;
    T const s_mag = norm(s_3D);
    T const sigma_yield = Y + K * alpha + (Y_inf - Y) * (1. - exp(-(delta * alpha)));
    T const f = (s_mag - sqrt_23 * sigma_yield) / val(mu);
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    Tens<T> R_zeta;
    T R_Ie, R_alpha;
    if (plastic) {
      Tens<T> const n_2D = mu * zeta / s_mag;
      T const dgam = sqrt_32 * (alpha - alpha_old);
      R_zeta = zeta - zeta_trial + 2. * dgam * Ie * n_2D;
      R_Ie = det(be_bar_3D) - 1.;
      R_alpha = f;
    } else {
      R_zeta = zeta - zeta_trial;
      R_Ie = Ie - Ie_trial;
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_zeta);
    this->set_scalar_R(1, R_Ie);
    this->set_scalar_R(2, R_alpha);
    return path;
  }
  Tens<T> cauchy(Global<T>& g) override {  // :328-337
    T const p = g.scalar_x(1);
    return this->dev_cauchy(g) - p * eye<T>(2);
  }
  Tens<T> dev_cauchy(Global<T>& g) override {  // :340-351
    T const mu = compute_mu(this->params[0], this->params[1]);
    Tens<T> const F = g.grad_vector_x(0) + eye<T>(2);
    Tens<T> const zeta = this->sym_tensor_xi(0);
    T const J = det(F);
    return mu * zeta / J;
  }
  T hydro_cauchy(Global<T>& g) override {  // :354-365
    T const kappa = compute_kappa(this->params[0], this->params[1]);
    Tens<T> const F = g.grad_vector_x(0) + eye<T>(2);
    T const J = det(F);
    return kappa / 2. * (J - 1. / J);
  }
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

// minitensor::polar_rotation (Trilinos MiniTensor_LinearAlgebra.t.h, third party, not under /root/reference): the
// rotation R of F = R U by Newton's iteration X <- (mu X + X^-T / mu) / 2 with Higham's 1-norm/inf-norm scaling
// ("Functions of Matrices", algorithm 8.20), differentiated through like any other arithmetic.  The converged R is
// unique, so the pin below does not depend on the iteration's details.
template <class T> T norm_1(Tens<T> const& A) {  // largest absolute column sum
  T best = abs(A(0, 0)) + abs(A(1, 0)) + abs(A(2, 0));
  for (int j = 1; j < 3; ++j) {
    T const s = abs(A(0, j)) + abs(A(1, j)) + abs(A(2, j));
    if (val(s) > val(best)) best = s;
  }
  return best;
}
template <class T> T norm_infinity(Tens<T> const& A) {  // largest absolute row sum
  T best = abs(A(0, 0)) + abs(A(0, 1)) + abs(A(0, 2));
  for (int i = 1; i < 3; ++i) {
    T const s = abs(A(i, 0)) + abs(A(i, 1)) + abs(A(i, 2));
    if (val(s) > val(best)) best = s;
  }
  return best;
}
template <class T> Tens<T> polar_rotation(Tens<T> const& A) {
  bool scale = true;
  double const tol_scale = 0.01;
  double const tol_conv = std::sqrt((double)A.dim) * 2.220446049250313e-16;  // sqrt(dimension) * machine epsilon
  Tens<T> X = A;
  double gamma = 2.0;
  for (int num_iter = 0; num_iter < 128; ++num_iter) {
    Tens<T> const Y = inverse(X);
    T mu = 1.0;
    if (scale) {
      mu = (norm_1(Y) * norm_infinity(Y)) / (norm_1(X) * norm_infinity(X));
      mu = sqrt(sqrt(mu));
    }
    Tens<T> const Z = 0.5 * (mu * X + transpose(Y) / mu);
    Tens<T> const D = Z - X;
    double const nD = val(norm(D));
    double const delta = nD / val(norm(Z));
    if (scale && delta < tol_scale) scale = false;
    bool const end_iter = nD <= std::sqrt(tol_conv) || (delta > 0.5 * gamma && !scale);
    X = Z;
    gamma = delta;
    if (end_iter) break;
  }
  return X;
}

// hypo_hill.cpp (TC SYM_TENSOR = unrotated Cauchy stress, alpha SCALAR; params E nu Y R00 R11 R22 R01 R02 R12 S D):
// hypoelastic rate form in the unrotated configuration, Hill yield function, Voce hardening
template <class T> struct HypoHill : Local<T> {
  HypoHill() { this->nres = 2; this->neq[0] = 6; this->neq[1] = 1; this->finish_layout(); }
  int num_params() const override { return 11; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 7; ++k) xi_pt[k] = 0.; }  // :123-131
  bool is_finite_deformation() const override { return true; }
  Tens<T> rotation(Global<T>& g) { return polar_rotation(g.grad_vector_x(0) + eye<T>()); }  // global_residual.hpp:302-305
  Tens<T> eval_d(Global<T>& g) {  // :134-139, hypo_kinematics.hpp:11-18
    Tens<T> const I = eye<T>();
    Tens<T> const F = g.grad_vector_x(0) + I;
    Tens<T> const F_prev = g.grad_vector_x_prev(0) + I;
    Tens<T> const R = polar_rotation(F);
    Tens<T> const L = (F - F_prev) * inverse(F);
    Tens<T> const D = 0.5 * (L + transpose(L));
    return transpose(R) * D * R;
  }
  int solve_nonlinear(Global<T>& g) override {  // :147-203
    if (std::is_same<T, double>::value) return 0;
    {
      double const E = val(this->params[0]), nu = val(this->params[1]);
      double const lambda = compute_lambda(E, nu), mu = compute_mu(E, nu);
      Tens<T> const I = eye<T>();
      Tens<T> const TC_old = this->sym_tensor_xi_prev(0);
      T const alpha_old = this->scalar_xi_prev(1);
      Tens<T> const d = eval_d(g);
      Tens<T> const TC = TC_old + (lambda * trace(d)) * I + (2. * mu) * d;
      this->set_sym_tensor_xi_val(0, TC);
      this->set_scalar_xi_val(1, val(alpha_old));
    }
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :211-289
    int path = ELASTIC_PATH;
    T const E = this->params[0], nu = this->params[1], Y = this->params[2];
    T const R00 = this->params[3], R11 = this->params[4], R22 = this->params[5];
    T const R01 = this->params[6], R02 = this->params[7], R12 = this->params[8];
    T const S = this->params[9], D = this->params[10];
    T const lambda = compute_lambda(E, nu);
    T const mu = compute_mu(E, nu);
    auto inv2 = [](T const& r) { return 1. / (r * r); };  // std::pow(r, -2)
    T hp[6];  // compute_hill_params (yield_functions.hpp:35-50)
    hp[0] = 0.5 * (inv2(R11) + inv2(R22) - inv2(R00));
    hp[1] = 0.5 * (inv2(R22) + inv2(R00) - inv2(R11));
    hp[2] = 0.5 * (inv2(R00) + inv2(R11) - inv2(R22));
    hp[3] = 1.5 * inv2(R12);
    hp[4] = 1.5 * inv2(R02);
    hp[5] = 1.5 * inv2(R01);
    Tens<T> const TC_old = this->sym_tensor_xi_prev(0);
    T const alpha_old = this->scalar_xi_prev(1);
    Tens<T> const TC = this->sym_tensor_xi(0);
    T const alpha = this->scalar_xi(1);
    T const d12 = TC(1, 1) - TC(2, 2), d20 = TC(2, 2) - TC(0, 0), d01 = TC(0, 0) - TC(1, 1);
    T const hill = sqrt(hp[0] * d12 * d12 + hp[1] * d20 * d20 + hp[2] * d01 * d01 +
                        2. * (hp[3] * TC(1, 2) * TC(1, 2) + hp[4] * TC(0, 2) * TC(0, 2) + hp[5] * TC(0, 1) * TC(0, 1)));
    T const sigma_yield = Y + S * (1. - exp(-(D * alpha)));
    T const f = (hill - sigma_yield) / val(mu);
    Tens<T> const I = eye<T>();
    Tens<T> const d = eval_d(g);
    Tens<T> R_TC = TC - TC_old - (lambda * trace(d)) * I - (2. * mu) * d;
    R_TC = R_TC / val(mu);
    T R_alpha;
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    if (plastic) {
      Tens<T> n;  // compute_hill_normal (:74-99)
      n(0, 0) = (hp[1] + hp[2]) * TC(0, 0) - hp[2] * TC(1, 1) - hp[1] * TC(2, 2);
      n(1, 1) = (hp[0] + hp[2]) * TC(1, 1) - hp[2] * TC(0, 0) - hp[0] * TC(2, 2);
      n(2, 2) = (hp[1] + hp[0]) * TC(2, 2) - hp[1] * TC(0, 0) - hp[0] * TC(1, 1);
      n(0, 1) = hp[5] * TC(0, 1); n(0, 2) = hp[4] * TC(0, 2); n(1, 2) = hp[3] * TC(1, 2);
      n(1, 0) = n(0, 1); n(2, 0) = n(0, 2); n(2, 1) = n(1, 2);
      n = n / hill;
      T const dgam = alpha - alpha_old;
      R_TC = R_TC + ((2. * mu * dgam) * n) / val(mu);
      R_alpha = f;
    } else {
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_TC);
    this->set_scalar_R(1, R_alpha);
    return path;
  }
  Tens<T> rotated_cauchy(Global<T>& g) {  // :292-298
    Tens<T> const TC = this->sym_tensor_xi(0);
    Tens<T> const R = rotation(g);
    return R * TC * transpose(R);
  }
  Tens<T> cauchy(Global<T>& g) override {  // :301-310
    T const p = g.scalar_x(1);
    return this->dev_cauchy(g) - p * eye<T>();
  }
  Tens<T> dev_cauchy(Global<T>& g) override { return dev(rotated_cauchy(g)); }  // :313-316
  T hydro_cauchy(Global<T>& g) override { return trace(rotated_cauchy(g)) / 3.; }  // :319-322
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

// hypo_hill_plane_strain.cpp (2-D: TC SYM_TENSOR (00,01,11) = in-plane unrotated Cauchy stress, alpha SCALAR, TC_zz
// SCALAR; params E nu Y S D R00 R11 R22 R01; R02 = R12 = 1): the hypoelastic rate form of hypo_hill on 2 x 2 kinematics
// with the out-of-plane stress as an extra unknown; residuals are not scaled by 1/mu here (unlike hypo_hill.cpp)
template <class T> struct HypoHillPlaneStrain : Local<T> {
  HypoHillPlaneStrain() { this->ndims = 2; this->nres = 3; this->neq[0] = 3; this->neq[1] = 1; this->neq[2] = 1; this->finish_layout(); }
  int num_params() const override { return 9; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 5; ++k) xi_pt[k] = 0.; }
  bool is_finite_deformation() const override { return true; }
  Tens<T> eval_d(Global<T>& g) const {  // :143-156
    Tens<T> const I = eye<T>(2);
    Tens<T> const F = g.grad_vector_x(0) + I;
    Tens<T> const F_prev = g.grad_vector_x_prev(0) + I;
    Tens<T> const Finv = inverse(F);
    Tens<T> const R = polar_rotation(F);
    Tens<T> const L = (F - F_prev) * Finv;
    Tens<T> const D = 0.5 * (L + transpose(L));
    return transpose(R) * D * R;
  }
  int solve_nonlinear(Global<T>& g) override {  // :163-224
    if (std::is_same<T, double>::value) return 0;
    {
      double const E = val(this->params[0]), nu = val(this->params[1]);
      double const lambda = compute_lambda(E, nu), mu = compute_mu(E, nu);
      Tens<T> const I = eye<T>(2);
      Tens<T> const TC_old = this->sym_tensor_xi_prev(0);
      T const alpha_old = this->scalar_xi_prev(1);
      T const TC_zz_old = this->scalar_xi_prev(2);
      Tens<T> const d = eval_d(g);
      Tens<T> const TC = TC_old + lambda * trace(d) * I + 2. * mu * d;
      T const TC_zz = TC_zz_old + lambda * trace(d);
      this->set_sym_tensor_xi_val(0, TC);
      this->set_scalar_xi_val(1, val(alpha_old));
      this->set_scalar_xi_val(2, val(TC_zz));
    }
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :232-329
    int path = ELASTIC_PATH;
    T const E = this->params[0], nu = this->params[1], Y = this->params[2], S = this->params[3], D = this->params[4];
    T const R00 = this->params[5], R11 = this->params[6], R22 = this->params[7], R01 = this->params[8];
    T const R02 = 1., R12 = 1.;
    T const lambda = compute_lambda(E, nu);
    T const mu = compute_mu(E, nu);
    auto inv2 = [](T const& r) { return 1. / (r * r); };
    T hp[6];
    hp[0] = 0.5 * (inv2(R11) + inv2(R22) - inv2(R00));
    hp[1] = 0.5 * (inv2(R22) + inv2(R00) - inv2(R11));
    hp[2] = 0.5 * (inv2(R00) + inv2(R11) - inv2(R22));
    hp[3] = 1.5 * inv2(R12);
    hp[4] = 1.5 * inv2(R02);
    hp[5] = 1.5 * inv2(R01);
    Tens<T> const TC_old = this->sym_tensor_xi_prev(0);
    T const alpha_old = this->scalar_xi_prev(1);
    T const TC_zz_old = this->scalar_xi_prev(2);
    Tens<T> const TC = this->sym_tensor_xi(0);
    T const alpha = this->scalar_xi(1);
    T const TC_zz = this->scalar_xi(2);
    Tens<T> s = TC;  // TC_3D
    s.dim = 3;
    s(2, 2) = TC_zz;
    T const d12 = s(1, 1) - s(2, 2), d20 = s(2, 2) - s(0, 0), d01 = s(0, 0) - s(1, 1);
    T const phi = sqrt(hp[0] * d12 * d12 + hp[1] * d20 * d20 + hp[2] * d01 * d01 +
                       2. * (hp[3] * s(1, 2) * s(1, 2) + hp[4] * s(0, 2) * s(0, 2) + hp[5] * s(0, 1) * s(0, 1)));
    T const sigma_yield = Y + S * (1. - exp(-(D * alpha)));
    T const f = (phi - sigma_yield) / val(mu);
    Tens<T> const I = eye<T>(2);
    Tens<T> const d = eval_d(g);
    Tens<T> R_TC = TC - TC_old - lambda * trace(d) * I - 2. * mu * d;
    T R_TC_zz = TC_zz - TC_zz_old - lambda * trace(d);
    T R_alpha;
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    if (plastic) {
      Tens<T> n;  // in-plane part of compute_hill_normal
      n.dim = 2;
      n(0, 0) = ((hp[1] + hp[2]) * s(0, 0) - hp[2] * s(1, 1) - hp[1] * s(2, 2)) / phi;
      n(1, 1) = ((hp[0] + hp[2]) * s(1, 1) - hp[2] * s(0, 0) - hp[0] * s(2, 2)) / phi;
      n(0, 1) = (hp[5] * s(0, 1)) / phi;
      n(1, 0) = n(0, 1);
      T const dgam = alpha - alpha_old;
      Tens<T> const dp_2D = dgam * n;
      T const dp_zz = -trace(dp_2D);
      R_TC = R_TC + 2. * mu * dp_2D;
      R_alpha = f;
      R_TC_zz = R_TC_zz + 2. * mu * dp_zz;
    } else {
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_TC);
    this->set_scalar_R(1, R_alpha);
    this->set_scalar_R(2, R_TC_zz);
    return path;
  }
  Tens<T> rotated_cauchy(Global<T>& g) {  // :332-341
    Tens<T> const F = g.grad_vector_x(0) + eye<T>(2);
    Tens<T> const TC = this->sym_tensor_xi(0);
    Tens<T> const R = polar_rotation(F);
    return R * TC * transpose(R);
  }
  Tens<T> cauchy(Global<T>& g) override { return this->dev_cauchy(g) - g.scalar_x(1) * eye<T>(2); }  // :343-352
  Tens<T> dev_cauchy(Global<T>& g) override { return rotated_cauchy(g) - this->hydro_cauchy(g) * eye<T>(2); }  // :355-362
  T hydro_cauchy(Global<T>& g) override {  // :365-369
    Tens<T> const RC = rotated_cauchy(g);
    return (trace(RC) + this->scalar_xi(2)) / 3.;
  }
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};


// ---------------------------------------------------------------------------
// Plane-stress family: local models that pair with `mechanics_plane_stress` (one global residual, u).  cauchy() is the
// in-plane Cauchy stress (sigma_zz = 0 is built into the models); pressure_scale_factor() = 0 is never used.
// ---------------------------------------------------------------------------
template <class T> static void hill_params_2d(std::vector<T> const& prm, T* hp) {  // compute_hill_params, yield_functions.hpp:35-50
  T const R00 = prm[5], R11 = prm[6], R22 = prm[7], R01 = prm[8];
  T const R02 = 1., R12 = 1.;
  auto inv2 = [](T const& r) { return 1. / (r * r); };
  hp[0] = 0.5 * (inv2(R11) + inv2(R22) - inv2(R00));
  hp[1] = 0.5 * (inv2(R22) + inv2(R00) - inv2(R11));
  hp[2] = 0.5 * (inv2(R00) + inv2(R11) - inv2(R22));
  hp[3] = 1.5 * inv2(R12);
  hp[4] = 1.5 * inv2(R02);
  hp[5] = 1.5 * inv2(R01);
}
template <class T> static T hill_value(Tens<T> const& s, T const* hp) {  // compute_hill_value, :53-72
  T const d12 = s(1, 1) - s(2, 2), d20 = s(2, 2) - s(0, 0), d01 = s(0, 0) - s(1, 1);
  return sqrt(hp[0] * d12 * d12 + hp[1] * d20 * d20 + hp[2] * d01 * d01 +
              2. * (hp[3] * s(1, 2) * s(1, 2) + hp[4] * s(0, 2) * s(0, 2) + hp[5] * s(0, 1) * s(0, 1)));
}
template <class T> static Tens<T> hill_normal_2d(Tens<T> const& s, T const* hp, T const& hill) {  // in-plane part of :75-98
  Tens<T> n;
  n.dim = 2;
  n(0, 0) = ((hp[1] + hp[2]) * s(0, 0) - hp[2] * s(1, 1) - hp[1] * s(2, 2)) / hill;
  n(1, 1) = ((hp[0] + hp[2]) * s(1, 1) - hp[2] * s(0, 0) - hp[0] * s(2, 2)) / hill;
  n(0, 1) = (hp[5] * s(0, 1)) / hill;
  n(1, 0) = n(0, 1);
  return n;
}
template <class T> static Tens<T> into_3d(Tens<T> const& t2) { Tens<T> t = t2; t.dim = 3; return t; }  // yield_functions.hpp:9-20

// small_hill_plane_stress.cpp (pstrain SYM_TENSOR (00,01,11), alpha SCALAR; params E nu Y S D R00 R11 R22 R01)
template <class T> struct SmallHillPlaneStress : Local<T> {
  SmallHillPlaneStress() { this->ndims = 2; this->nres = 2; this->neq[0] = 3; this->neq[1] = 1; this->finish_layout(); }
  int num_params() const override { return 9; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 4; ++k) xi_pt[k] = 0.; }  // :112-124
  bool is_finite_deformation() const override { return false; }
  int solve_nonlinear(Global<T>& g) override {  // :131-184
    if (std::is_same<T, double>::value) return 0;
    this->set_sym_tensor_xi_val(0, this->sym_tensor_xi_prev(0));
    this->set_scalar_xi_val(1, val(this->scalar_xi_prev(1)));
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :192-275
    int path = ELASTIC_PATH;
    T const E = this->params[0], nu = this->params[1], Y = this->params[2], S = this->params[3], D = this->params[4];
    T const mu = compute_mu(E, nu);
    T hp[6];
    hill_params_2d(this->params, hp);
    Tens<T> const pstrain_old = this->sym_tensor_xi_prev(0);
    T const alpha_old = this->scalar_xi_prev(1);
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    T const alpha = this->scalar_xi(1);
    Tens<T> const sigma_3D = into_3d(this->cauchy(g));
    T const hill = hill_value(sigma_3D, hp);
    T const sigma_yield = Y + S * (1. - exp(-(D * alpha)));
    T const f = (hill - sigma_yield) / val(mu);
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    Tens<T> R_pstrain;
    T R_alpha;
    if (plastic) {
      Tens<T> const n_2D = hill_normal_2d(sigma_3D, hp, hill);
      T const dgam = alpha - alpha_old;
      R_pstrain = pstrain - pstrain_old - dgam * n_2D;
      R_alpha = f;
    } else {
      R_pstrain = pstrain - pstrain_old;
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_pstrain);
    this->set_scalar_R(1, R_alpha);
    return path;
  }
  T epsilon_zz(Global<T>& g) {  // :318-329
    T const E = this->params[0], nu = this->params[1];
    T const mu = compute_mu(E, nu), lambda = compute_lambda(E, nu);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const epsilon = 0.5 * (grad_u + transpose(grad_u));
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    return -(lambda * trace(epsilon) + 2. * mu * trace(pstrain)) / (lambda + 2. * mu);
  }
  Tens<T> cauchy(Global<T>& g) override {  // :278-293
    T const E = this->params[0], nu = this->params[1];
    T const mu = compute_mu(E, nu), lambda = compute_lambda(E, nu);
    Tens<T> const I = eye<T>(2);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const epsilon = 0.5 * (grad_u + transpose(grad_u));
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    T const epsilon_kk = trace(epsilon) + epsilon_zz(g);
    return lambda * epsilon_kk * I + 2. * mu * (epsilon - pstrain);
  }
  T hydro_cauchy(Global<T>& g) override { return trace(cauchy(g)) / 3.; }                                 // :305-309
  Tens<T> dev_cauchy(Global<T>& g) override { return cauchy(g) - hydro_cauchy(g) * eye<T>(2); }          // :296-302
  T pressure_scale_factor() override { return 0.; }
};

// hyper_J2_plane_stress.cpp (zeta SYM_TENSOR (00,01,11), Ie, lambda_z, alpha SCALAR; params E nu Y S D A n K)
template <class T> struct HyperJ2PlaneStress : Local<T> {
  HyperJ2PlaneStress() {
    this->ndims = 2; this->nres = 4; this->neq[0] = 3; this->neq[1] = this->neq[2] = this->neq[3] = 1;
    this->z_stretch_idx = 2;
    this->finish_layout();
  }
  int num_params() const override { return 8; }
  void init_variables(double* xi_pt) const override {  // :121-138
    for (int k = 0; k < 6; ++k) xi_pt[k] = 0.;
    xi_pt[3] = 1.;
    xi_pt[4] = 1.;
  }
  bool is_finite_deformation() const override { return true; }
  // eval_be_bar_plane_stress (:141-169)
  void be_bar_trial(Global<T>& g, Tens<T> const& zeta_2D, T const& Ie, T const& lambda_z_prev, T const& lambda_z, T& J_2D,
                    Tens<T>& be_bar) const {
    Tens<T> const I_2D = eye<T>(2);
    Tens<T> const I = eye<T>(3);
    Tens<T> const F_2D = g.grad_vector_x(0) + I_2D;
    J_2D = det(F_2D);
    Tens<T> const F_prev_2D = g.grad_vector_x_prev(0) + I_2D;
    Tens<T> F_3D = into_3d(F_2D), F_prev_3D = into_3d(F_prev_2D);
    F_3D(2, 2) = lambda_z;
    F_prev_3D(2, 2) = lambda_z_prev;
    Tens<T> const rF = F_3D * inverse(F_prev_3D);
    T const det_rF = det(rF);
    T const det_rF_13 = cbrt(det_rF);
    Tens<T> const rF_bar = rF / det_rF_13;
    Tens<T> const rF_barT = transpose(rF_bar);
    Tens<T> zeta_3D = into_3d(zeta_2D);
    zeta_3D(2, 2) = -(zeta_2D(0, 0) + zeta_2D(1, 1));
    be_bar = rF_bar * (zeta_3D + Ie * I) * rF_barT;
  }
  static Tens<T> in_plane(Tens<T> const& t3) {  // extract_2D_tensor_from_3D, yield_functions.hpp:22-33
    Tens<T> t;
    t.dim = 2;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) t(i, j) = t3(i, j);
    return t;
  }
  int solve_nonlinear(Global<T>& g) override {  // :176-238
    if (std::is_same<T, double>::value) return 0;
    {
      T J_2D;
      Tens<T> bt;
      be_bar_trial(g, this->sym_tensor_xi_prev(0), this->scalar_xi_prev(1), this->scalar_xi_prev(2), this->scalar_xi(2), J_2D, bt);
      T const Ie_trial = (bt(0, 0) + bt(1, 1) + bt(2, 2)) / 3.;
      Tens<T> const zeta_trial_3D = bt - Ie_trial * eye<T>(3);
      this->set_sym_tensor_xi_val(0, in_plane(zeta_trial_3D));
      this->set_scalar_xi_val(1, val(Ie_trial));
    }
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :246-358
    int path = ELASTIC_PATH;
    double const sqrt_23 = std::sqrt(2. / 3.), sqrt_32 = std::sqrt(3. / 2.);
    T const E = this->params[0], nu = this->params[1], Y = this->params[2], S = this->params[3], D = this->params[4];
    T const A = this->params[5], n = this->params[6], K = this->params[7];
    T const mu = compute_mu(E, nu), kappa = compute_kappa(E, nu);
    Tens<T> const zeta_old = this->sym_tensor_xi_prev(0);
    T const Ie_old = this->scalar_xi_prev(1), lambda_z_old = this->scalar_xi_prev(2), alpha_old = this->scalar_xi_prev(3);
    Tens<T> const zeta = this->sym_tensor_xi(0);
    T const Ie = this->scalar_xi(1), lambda_z = this->scalar_xi(2), alpha = this->scalar_xi(3);
    Tens<T> const I = eye<T>(3);
    T J_2D;
    Tens<T> bt;
    be_bar_trial(g, zeta_old, Ie_old, lambda_z_old, lambda_z, J_2D, bt);
    T const Ie_trial = (bt(0, 0) + bt(1, 1) + bt(2, 2)) / 3.;
    Tens<T> const zeta_trial_2D = in_plane(bt - Ie_trial * I);
    Tens<T> zeta_3D = into_3d(zeta);
    T const zeta_zz = -(zeta(0, 0) + zeta(1, 1));
    zeta_3D(2, 2) = zeta_zz;
    Tens<T> const be_bar = zeta_3D + Ie * I;
    Tens<T> const s = mu * zeta_3D;
    T const s_mag = norm(s);
    double const power_law_offset = 1e-12;
    T const sigma_yield = Y + S * (1. - exp(-(D * alpha))) + A * pow(alpha + power_law_offset, n) + K * alpha;
    T const f = (s_mag - sqrt_23 * sigma_yield) / val(mu);
    T const mat_factor = kappa / (2. * mu);
    T const R_lambda_z = lambda_z - sqrt((1. - zeta_zz / mat_factor) / (J_2D * J_2D));
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    Tens<T> R_zeta;
    T R_Ie, R_alpha;
    if (plastic) {
      Tens<T> const n_2D = mu * zeta / s_mag;
      T const dgam = sqrt_32 * (alpha - alpha_old);
      R_zeta = zeta - zeta_trial_2D + 2. * dgam * Ie * n_2D;
      R_Ie = det(be_bar) - 1.;
      R_alpha = f;
    } else {
      R_zeta = zeta - zeta_trial_2D;
      R_Ie = Ie - Ie_trial;
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_zeta);
    this->set_scalar_R(1, R_Ie);
    this->set_scalar_R(2, R_lambda_z);
    this->set_scalar_R(3, R_alpha);
    return path;
  }
  T jac(Global<T>& g) { return det(g.grad_vector_x(0) + eye<T>(2)) * this->scalar_xi(2); }
  Tens<T> cauchy(Global<T>& g) override {  // :361-374
    T const mu = compute_mu(this->params[0], this->params[1]), kappa = compute_kappa(this->params[0], this->params[1]);
    T const J = jac(g);
    return mu * this->sym_tensor_xi(0) / J + (kappa / 2. * (J - 1. / J)) * eye<T>(2);
  }
  Tens<T> dev_cauchy(Global<T>& g) override {  // :377-389
    T const mu = compute_mu(this->params[0], this->params[1]);
    return mu * this->sym_tensor_xi(0) / jac(g);
  }
  T hydro_cauchy(Global<T>& g) override {  // :392-403
    T const kappa = compute_kappa(this->params[0], this->params[1]);
    T const J = jac(g);
    return kappa / 2. * (J - 1. / J);
  }
  T pressure_scale_factor() override { return 0.; }
};


// hypo_hill_plane_stress.cpp (TC SYM_TENSOR (00,01,11), alpha, lambda_z SCALAR; params E nu Y S D R00 R11 R22 R01 Q00 Q01
// Q10 Q11): the material axes Q enter the rate of deformation (:164-177) and the rotated stress (:378-388); the TC rows of
// the plastic residual are divided by val(mu) on the unforced path only (:303), as the reference does.
template <class T> struct HypoHillPlaneStress : Local<T> {
  HypoHillPlaneStress() {
    this->ndims = 2; this->nres = 3; this->neq[0] = 3; this->neq[1] = 1; this->neq[2] = 1;
    this->z_stretch_idx = 2;
    this->finish_layout();
  }
  int num_params() const override { return 13; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 5; ++k) xi_pt[k] = 0.; xi_pt[4] = 1.; }  // :138-152
  bool is_finite_deformation() const override { return true; }
  Tens<T> compute_Q() const {  // :155-162
    Tens<T> Q;
    Q.dim = 2;
    Q(0, 0) = this->params[9]; Q(0, 1) = this->params[10]; Q(1, 0) = this->params[11]; Q(1, 1) = this->params[12];
    return Q;
  }
  Tens<T> eval_d(Global<T>& g, Tens<T> const& Q) const {  // :164-177
    Tens<T> const I = eye<T>(2);
    Tens<T> const F = g.grad_vector_x(0) + I;
    Tens<T> const F_prev = g.grad_vector_x_prev(0) + I;
    Tens<T> const Finv = inverse(F);
    Tens<T> const R = polar_rotation(F);
    Tens<T> const L = (F - F_prev) * Finv;
    Tens<T> const D = 0.5 * (L + transpose(L));
    return transpose(Q) * transpose(R) * D * R * Q;
  }
  int solve_nonlinear(Global<T>& g) override {  // :184-248
    if (std::is_same<T, double>::value) return 0;
    {
      double const E = val(this->params[0]), nu = val(this->params[1]);
      double const lambda = compute_lambda(E, nu), mu = compute_mu(E, nu);
      Tens<T> const I = eye<T>(2);
      Tens<T> const TC_old = this->sym_tensor_xi_prev(0);
      T const lambda_z_old = this->scalar_xi_prev(2);
      Tens<T> const d = eval_d(g, compute_Q());
      T const d_zz = -lambda * trace(d) / (lambda + 2. * mu);
      Tens<T> const TC = TC_old + lambda * (trace(d) + d_zz) * I + 2. * mu * d;
      T const lambda_z = lambda_z_old / (1. - d_zz);
      this->set_sym_tensor_xi_val(0, TC);
      this->set_scalar_xi_val(1, val(this->scalar_xi_prev(1)));
      this->set_scalar_xi_val(2, val(lambda_z));
    }
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :256-375
    int path = ELASTIC_PATH;
    T const E = this->params[0], nu = this->params[1], Y = this->params[2], S = this->params[3], D = this->params[4];
    T const lambda = compute_lambda(E, nu), mu = compute_mu(E, nu);
    Tens<T> const TC_old = this->sym_tensor_xi_prev(0);
    T const alpha_old = this->scalar_xi_prev(1), lambda_z_old = this->scalar_xi_prev(2);
    Tens<T> const TC = this->sym_tensor_xi(0);
    T const alpha = this->scalar_xi(1), lambda_z = this->scalar_xi(2);
    Tens<T> const TC_3D = into_3d(TC);
    T hp[6];
    hill_params_2d(this->params, hp);
    T const phi = hill_value(TC_3D, hp);
    T const sigma_yield = Y + S * (1. - exp(-(D * alpha)));
    T const f = (phi - sigma_yield) / val(mu);
    Tens<T> const I = eye<T>(2);
    Tens<T> const d = eval_d(g, compute_Q());
    T const d_zz = -lambda * trace(d) / (lambda + 2. * mu);
    Tens<T> R_TC = TC - TC_old - lambda * (trace(d) + d_zz) * I - 2. * mu * d;
    T R_alpha, R_lambda_z;
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    if (plastic) {
      Tens<T> const n_2D = hill_normal_2d(TC_3D, hp, phi);
      T const dgam = alpha - alpha_old;
      Tens<T> const dp_2D = dgam * n_2D;
      T const dp_zz = -(dp_2D(0, 0) + dp_2D(1, 1));
      T const corr_dp_zz = 2. * mu * dp_zz / (2. * mu + lambda);  // correction from the return map
      R_TC(0, 0) = R_TC(0, 0) + (2. * mu * dp_2D(0, 0) - lambda * corr_dp_zz);
      R_TC(1, 1) = R_TC(1, 1) + (2. * mu * dp_2D(1, 1) - lambda * corr_dp_zz);
      R_TC(0, 1) = R_TC(0, 1) + 2. * mu * dp_2D(0, 1);
      if (!force_path) R_TC = R_TC / val(mu);  // :303 (only there)
      R_alpha = f;
      R_lambda_z = lambda_z - lambda_z_old / (1. - (d_zz + corr_dp_zz));
    } else {
      R_alpha = alpha - alpha_old;
      R_lambda_z = lambda_z - lambda_z_old / (1. - d_zz);
    }
    this->set_sym_tensor_R(0, R_TC);
    this->set_scalar_R(1, R_alpha);
    this->set_scalar_R(2, R_lambda_z);
    return path;
  }
  Tens<T> rotated_cauchy(Global<T>& g) {  // :378-388
    Tens<T> const Q = compute_Q();
    Tens<T> const F = g.grad_vector_x(0) + eye<T>(2);
    Tens<T> const TC = this->sym_tensor_xi(0);
    Tens<T> const R = polar_rotation(F);
    return R * Q * TC * transpose(Q) * transpose(R);
  }
  Tens<T> cauchy(Global<T>& g) override { return rotated_cauchy(g); }                                        // :391-393
  Tens<T> dev_cauchy(Global<T>& g) override { return rotated_cauchy(g) - hydro_cauchy(g) * eye<T>(2); }   // :396-400
  T hydro_cauchy(Global<T>& g) override { return trace(rotated_cauchy(g)) / 3.; }                          // :403-405
  T pressure_scale_factor() override { return 0.; }
};


// ---------------------------------------------------------------------------
// Hosford's isotropic yield function on the principal stresses (small_hosford.cpp:228-265, hypo_hosford.cpp:264-301):
// phi = vm (1/2 sum |(s_i - s_j)/vm|^a)^(1/a), scaled by the von Mises stress so that the powers stay bounded, and its
// normal through the eigen-dyads.
// ---------------------------------------------------------------------------
template <class T> static void hosford_phi_and_normal(Tens<T> const& sigma, T const& vm_stress, T const& a, T& phi, Tens<T>& n) {
  Tens<T> V, D;
  eig_spd_cos(sigma, V, D);
  T const e0 = D(0, 0) / vm_stress, e1 = D(1, 1) / vm_stress, e2 = D(2, 2) / vm_stress;
  phi = vm_stress * pow(0.5 * (pow(abs(e0 - e1), a) + pow(abs(e1 - e2), a) + pow(abs(e2 - e0), a)), 1. / a);
  T const q0 = D(0, 0) / phi, q1 = D(1, 1) / phi, q2 = D(2, 2) / phi;
  T const d01 = q0 - q1, d12 = q1 - q2, d20 = q2 - q0;
  T const f01 = d01 * pow(abs(d01), a - 2.), f12 = d12 * pow(abs(d12), a - 2.), f20 = d20 * pow(abs(d20), a - 2.);
  n = 0.5 * ((f01 - f20) * dyad_col(V, 0) + (f12 - f01) * dyad_col(V, 1) + (f20 - f12) * dyad_col(V, 2));
}

// small_hosford.cpp (pstrain SYM_TENSOR, alpha SCALAR; params E nu Y a K S D)
template <class T> struct SmallHosford : Local<T> {
  SmallHosford() { this->nres = 2; this->neq[0] = 6; this->neq[1] = 1; this->finish_layout(); }
  int num_params() const override { return 7; }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 7; ++k) xi_pt[k] = 0.; }  // :117-129
  bool is_finite_deformation() const override { return false; }
  int solve_nonlinear(Global<T>& g) override {  // :136-224
    if (std::is_same<T, double>::value) return 0;
    this->set_sym_tensor_xi_val(0, this->sym_tensor_xi_prev(0));
    this->set_scalar_xi_val(1, val(this->scalar_xi_prev(1)));
    return this->newton_ls(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :268-340
    int path = ELASTIC_PATH;
    T const E = this->params[0], nu = this->params[1], Y = this->params[2], a = this->params[3], K = this->params[4];
    T const S = this->params[5], D = this->params[6];
    T const mu = compute_mu(E, nu);
    Tens<T> const pstrain_old = this->sym_tensor_xi_prev(0);
    T const alpha_old = this->scalar_xi_prev(1);
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    T const alpha = this->scalar_xi(1);
    T phi = 0.;
    Tens<T> n;
    T const vm_stress = std::sqrt(3. / 2.) * norm(this->dev_cauchy(g));
    hosford_phi_and_normal(this->cauchy(g), vm_stress, a, phi, n);
    T const flow_stress = Y + K * alpha + S * (1. - exp(-(D * alpha)));
    T const f = (phi - flow_stress) / (2. * val(mu));
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    Tens<T> R_pstrain;
    T R_alpha;
    if (plastic) {
      T const dgam = alpha - alpha_old;
      R_pstrain = pstrain - pstrain_old - dgam * n;
      R_alpha = f;
    } else {
      R_pstrain = pstrain - pstrain_old;
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_pstrain);
    this->set_scalar_R(1, R_alpha);
    return path;
  }
  Tens<T> cauchy(Global<T>& g) override { return this->dev_cauchy(g) - g.scalar_x(1) * eye<T>(); }  // :343-353
  Tens<T> dev_cauchy(Global<T>& g) override {  // :356-368
    T const mu = compute_mu(this->params[0], this->params[1]);
    Tens<T> const pstrain = this->sym_tensor_xi(0);
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    Tens<T> const dev_eps = eps - (trace(eps) / 3.) * eye<T>();
    return (2. * mu) * (dev_eps - pstrain);
  }
  T hydro_cauchy(Global<T>& g) override {  // :371-378
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const eps = 0.5 * (grad_u + transpose(grad_u));
    return compute_kappa(this->params[0], this->params[1]) * trace(eps);
  }
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

// the hypoelastic base of hypo_hosford.cpp / hypo_barlat.cpp: unrotated Cauchy stress TC and alpha, the stress measures
// of hypo_hill.cpp, the elastic predictor as the initial guess
template <class T> struct HypoBase : Local<T> {
  HypoBase() { this->nres = 2; this->neq[0] = 6; this->neq[1] = 1; this->finish_layout(); }
  void init_variables(double* xi_pt) const override { for (int k = 0; k < 7; ++k) xi_pt[k] = 0.; }
  bool is_finite_deformation() const override { return true; }
  Tens<T> eval_d(Global<T>& g) {  // hypo_kinematics.hpp:11-18 (the material axes Q of hypo_barlat.cpp default to I)
    Tens<T> const I = eye<T>();
    Tens<T> const F = g.grad_vector_x(0) + I;
    Tens<T> const F_prev = g.grad_vector_x_prev(0) + I;
    Tens<T> const R = polar_rotation(F);
    Tens<T> const L = (F - F_prev) * inverse(F);
    Tens<T> const D = 0.5 * (L + transpose(L));
    return transpose(R) * D * R;
  }
  int solve_nonlinear(Global<T>& g) override {  // hypo_hosford.cpp:160-260, hypo_barlat.cpp:327-438
    if (std::is_same<T, double>::value) return 0;
    {
      double const E = val(this->params[0]), nu = val(this->params[1]);
      double const lambda = compute_lambda(E, nu), mu = compute_mu(E, nu);
      Tens<T> const d = eval_d(g);
      Tens<T> const TC = this->sym_tensor_xi_prev(0) + (lambda * trace(d)) * eye<T>() + (2. * mu) * d;
      this->set_sym_tensor_xi_val(0, TC);
      this->set_scalar_xi_val(1, val(this->scalar_xi_prev(1)));
    }
    return this->newton_ls(g);
  }
  // the residual shared by both models once phi (and, on the plastic branch, the normal) is known
  template <class Normal>
  int finish(Global<T>& g, T const& phi, T const& flow_stress, Normal&& normal, bool force_path, int path_in) {
    int path;
    T const lambda = compute_lambda(this->params[0], this->params[1]), mu = compute_mu(this->params[0], this->params[1]);
    T const alpha_old = this->scalar_xi_prev(1), alpha = this->scalar_xi(1);
    Tens<T> const TC = this->sym_tensor_xi(0);
    T const scale_factor = 2. * mu;
    T const f = (phi - flow_stress) / scale_factor;
    Tens<T> const d = eval_d(g);
    Tens<T> R_TC = (TC - this->sym_tensor_xi_prev(0) - (lambda * trace(d)) * eye<T>() - (2. * mu) * d) / scale_factor;
    T R_alpha;
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    if (plastic) {
      T const dgam = alpha - alpha_old;
      R_TC = R_TC + dgam * normal();  // the scale factor of R_TC removes the 2 mu multiplier
      R_alpha = f;
    } else {
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_TC);
    this->set_scalar_R(1, R_alpha);
    return path;
  }
  Tens<T> rotated_cauchy(Global<T>& g) {
    Tens<T> const R = polar_rotation(g.grad_vector_x(0) + eye<T>());
    return R * this->sym_tensor_xi(0) * transpose(R);
  }
  Tens<T> cauchy(Global<T>& g) override { return this->dev_cauchy(g) - g.scalar_x(1) * eye<T>(); }
  Tens<T> dev_cauchy(Global<T>& g) override { return dev(rotated_cauchy(g)); }
  T hydro_cauchy(Global<T>& g) override { return trace(rotated_cauchy(g)) / 3.; }
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

// hypo_hosford.cpp (params E nu Y a K S D; the flow stress has no K term, :326)
template <class T> struct HypoHosford : HypoBase<T> {
  int num_params() const override { return 7; }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :304-378
    T const Y = this->params[2], a = this->params[3], S = this->params[5], D = this->params[6];
    Tens<T> const TC = this->sym_tensor_xi(0);
    T phi = 0.;
    Tens<T> n;
    T const vm_stress = std::sqrt(3. / 2.) * norm(dev(TC));
    hosford_phi_and_normal(TC, vm_stress, a, phi, n);
    T const flow_stress = Y + S * (1. - exp(-(D * this->scalar_xi(1))));
    return this->finish(g, phi, flow_stress, [&]() { return n; }, force_path, path_in);
  }
};

// hypo_barlat.cpp: Barlat's Yld2004-18p on two linear transformations of the stress (yield_functions.hpp:101-386);
// params E nu Y a K S D sp_01 .. sp_55 dp_01 .. dp_55
template <class T> struct HypoBarlat : HypoBase<T> {
  int num_params() const override { return 25; }
  static void transform(T const* q, Tens<T> const& sigma, T L[6][6], Tens<T>& s) {  // unflatten_barlat_params :162-189, L * stress
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) L[i][j] = 0.;
    L[0][0] = (q[0] + q[1]) / 3.;       L[0][1] = (-2. * q[0] + q[1]) / 3.; L[0][2] = (q[0] - 2. * q[1]) / 3.;
    L[1][0] = (-2. * q[2] + q[3]) / 3.; L[1][1] = (q[2] + q[3]) / 3.;       L[1][2] = (q[2] - 2. * q[3]) / 3.;
    L[2][0] = (-2. * q[4] + q[5]) / 3.; L[2][1] = (q[4] - 2. * q[5]) / 3.;  L[2][2] = (q[4] + q[5]) / 3.;
    L[3][3] = q[6]; L[4][4] = q[7]; L[5][5] = q[8];
    T v[6] = {sigma(0, 0), sigma(1, 1), sigma(2, 2), sigma(0, 1), sigma(1, 2), sigma(2, 0)}, w[6];  // flatten_stress :128-141
    for (int i = 0; i < 6; ++i) { w[i] = L[i][0] * v[0]; for (int j = 1; j < 6; ++j) w[i] += L[i][j] * v[j]; }
    s(0, 0) = w[0]; s(1, 1) = w[1]; s(2, 2) = w[2];
    s(0, 1) = s(1, 0) = w[3]; s(1, 2) = s(2, 1) = w[4]; s(0, 2) = s(2, 0) = w[5];
  }
  static void flatten(Tens<T> const& t, T* v) { v[0] = t(0, 0); v[1] = t(1, 1); v[2] = t(2, 2); v[3] = t(0, 1); v[4] = t(1, 2); v[5] = t(2, 0); }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :446-555
    T const Y = this->params[2], a = this->params[3], K = this->params[4], S = this->params[5], D = this->params[6];
    T const* sp = &this->params[7];
    T const* dp = &this->params[16];
    Tens<T> const TC = this->sym_tensor_xi(0);
    T const alpha = this->scalar_xi(1);
    // evaluate_barlat_phi (yield_functions.hpp:322-366)
    double const vm_phi = std::sqrt(3. / 2.) * val(norm(dev(TC)));
    T Ls[6][6], Ld[6][6];
    Tens<T> ss, sd, Vs, Ds, Vd, Dd;
    transform(sp, TC, Ls, ss);
    transform(dp, TC, Ld, sd);
    eig_spd_cos(ss, Vs, Ds);
    eig_spd_cos(sd, Vd, Dd);
    T sum = 0.;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) sum += pow(abs(Ds(i, i) / vm_phi - Dd(j, j) / vm_phi), a);
    sum = 0.25 * sum;
    T const phi = vm_phi * exp((1.0 / a) * log(sum));
    T const flow_stress = Y + K * alpha + S * (1. - exp(-(D * alpha)));
    auto normal = [&]() {  // evaluate_barlat_normal / compute_barlat_normal (:293-320, :369-384)
      Tens<T> sp_normal, dp_normal;
      for (int k = 0; k < 3; ++k) {
        T ms = 0., md = 0.;
        for (int j = 0; j < 3; ++j) {
          T const ds = Ds(k, k) / phi - Dd(j, j) / phi;  // sp multiplier (:230-246)
          ms += ds * pow(abs(ds), a - 2.);
          T const dd = Ds(j, j) / phi - Dd(k, k) / phi;  // dp multiplier (:258-274)
          md += -dd * pow(abs(dd), a - 2.);
        }
        sp_normal = sp_normal + (0.25 * ms) * dyad_col(Vs, k);
        dp_normal = dp_normal + (0.25 * md) * dyad_col(Vd, k);
      }
      T vs[6], vd[6], w[6];
      flatten(sp_normal, vs);
      flatten(dp_normal, vd);
      for (int i = 0; i < 6; ++i) {
        w[i] = 0.;
        for (int j = 0; j < 6; ++j) w[i] += Ls[i][j] * vs[j];
        for (int j = 0; j < 6; ++j) w[i] += Ld[i][j] * vd[j];
      }
      Tens<T> n;
      n(0, 0) = w[0]; n(1, 1) = w[1]; n(2, 2) = w[2];
      n(0, 1) = n(1, 0) = w[3]; n(1, 2) = n(2, 1) = w[4]; n(0, 2) = n(2, 0) = w[5];
      return n;
    };
    return this->finish(g, phi, flow_stress, normal, force_path, path_in);
  }
};

// hyper_J2.cpp (zeta SYM_TENSOR, Ie SCALAR, alpha SCALAR; params E nu Y S D A n K)
template <class T> struct HyperJ2 : Local<T> {
  HyperJ2() { this->nres = 3; this->neq[0] = 6; this->neq[1] = 1; this->neq[2] = 1; this->finish_layout(); }
  int num_params() const override { return 8; }
  void init_variables(double* xi_pt) const override {  // :119-134
    for (int k = 0; k < 8; ++k) xi_pt[k] = 0.;
    xi_pt[6] = 1.0;
  }
  bool is_finite_deformation() const override { return true; }
  Tens<T> eval_be_bar(Global<T>& g, Tens<T> const& zeta, T const& Ie) {  // :137-154
    Tens<T> const I = eye<T>();
    Tens<T> const grad_u = g.grad_vector_x(0);
    Tens<T> const grad_u_prev = g.grad_vector_x_prev(0);
    Tens<T> const F = grad_u + I;
    Tens<T> const F_prev = grad_u_prev + I;
    Tens<T> const rF = F * inverse(F_prev);
    T const det_rF = det(rF);
    T const det_rF_13 = cbrt(det_rF);
    Tens<T> const rF_bar = rF / det_rF_13;
    Tens<T> const rF_barT = transpose(rF_bar);
    return rF_bar * (zeta + Ie * I) * rF_barT;
  }
  int solve_nonlinear(Global<T>& g) override {  // :162-218
    if (std::is_same<T, double>::value) return 0;
    {
      Tens<T> const zeta_old = this->sym_tensor_xi_prev(0);
      T const Ie_old = this->scalar_xi_prev(1);
      T const alpha_old = this->scalar_xi_prev(2);
      Tens<T> const be_bar_trial = eval_be_bar(g, zeta_old, Ie_old);
      Tens<T> const zeta = dev(be_bar_trial);
      T const Ie = trace(be_bar_trial) / 3.;
      this->set_sym_tensor_xi_val(0, zeta);
      this->set_scalar_xi_val(1, val(Ie));
      this->set_scalar_xi_val(2, val(alpha_old));
    }
    return this->newton(g);
  }
  int evaluate(Global<T>& g, bool force_path, int path_in) override {  // :226-314
    int path = ELASTIC_PATH;
    double const sqrt_23 = std::sqrt(2. / 3.);
    double const sqrt_32 = std::sqrt(3. / 2.);
    T const E = this->params[0];
    T const nu = this->params[1];
    T const Y = this->params[2];
    T const S = this->params[3];
    T const D = this->params[4];
    T const A = this->params[5];
    T const nexp = this->params[6];
    T const K = this->params[7];
    T const mu = compute_mu(E, nu);
    Tens<T> const zeta_old = this->sym_tensor_xi_prev(0);
    T const Ie_old = this->scalar_xi_prev(1);
    T const alpha_old = this->scalar_xi_prev(2);
    Tens<T> const zeta = this->sym_tensor_xi(0);
    T const Ie = this->scalar_xi(1);
    T const alpha = this->scalar_xi(2);
    Tens<T> const I = eye<T>();
    Tens<T> const be_bar_trial = eval_be_bar(g, zeta_old, Ie_old);
    Tens<T> const s = mu * zeta;
    T const s_mag = norm(s);
    double const power_law_offset = 1e-12;
    T const sigma_yield = Y + S * (1. - exp(-D * alpha)) + A * pow(alpha + power_law_offset, nexp) + K * alpha;
    T const f = (s_mag - sqrt_23 * sigma_yield) / val(mu);
    Tens<T> R_zeta;
    T R_Ie, R_alpha;
    bool plastic;
    if (!force_path) {
      plastic = (f > this->abs_tol || abs(val(f)) < this->abs_tol);
      path = plastic ? PLASTIC_PATH : ELASTIC_PATH;
    } else {
      path = path_in;
      plastic = (path == PLASTIC_PATH);
    }
    if (plastic) {
      Tens<T> const n = s / s_mag;
      T const dgam = sqrt_32 * (alpha - alpha_old);
      R_zeta = zeta - dev(be_bar_trial) + ((2. * dgam) * Ie) * n;
      R_Ie = det(zeta + Ie * I) - 1.;
      R_alpha = f;
    } else {
      R_zeta = zeta - dev(be_bar_trial);
      R_Ie = Ie - trace(be_bar_trial) / 3.;
      R_alpha = alpha - alpha_old;
    }
    this->set_sym_tensor_R(0, R_zeta);
    this->set_scalar_R(1, R_Ie);
    this->set_scalar_R(2, R_alpha);
    return path;
  }
  Tens<T> cauchy(Global<T>& g) override {  // :316-324
    T const p = g.scalar_x(1);
    Tens<T> const I = eye<T>();
    Tens<T> const dev_sigma = this->dev_cauchy(g);
    return dev_sigma - p * I;
  }
  Tens<T> dev_cauchy(Global<T>& g) override {  // :327-338
    T const mu = compute_mu(this->params[0], this->params[1]);
    Tens<T> const I = eye<T>();
    Tens<T> const F = g.grad_vector_x(0) + I;
    Tens<T> const zeta = this->sym_tensor_xi(0);
    T const J = det(F);
    return (mu * zeta) / J;
  }
  T hydro_cauchy(Global<T>& g) override {  // :341-352
    T const kappa = compute_kappa(this->params[0], this->params[1]);
    Tens<T> const I = eye<T>();
    Tens<T> const F = g.grad_vector_x(0) + I;
    T const J = det(F);
    return kappa / 2. * (J - 1. / J);
  }
  T pressure_scale_factor() override { return compute_kappa(this->params[0], this->params[1]); }
};

template <class T> void Global<T>::evaluate(Local<T>& local, double w, double dv, int ip_set) {
  if (nres == 1) {  // MechanicsPlaneStress::evaluate, mechanics_plane_stress.cpp:47-95: momentum balance only
    Tens<T> stress = local.cauchy(*this);
    if (local.is_finite_deformation()) {  // :66-82
      Tens<T> const F_invT = transpose(inverse(F));
      T const z_stretch = local.xi[local.off[local.z_stretch_idx]];
      stress = z_stretch * det_F * stress * F_invT;
    }
    for (int n = 0; n < nn; ++n)
      for (int i = 0; i < ndims; ++i)
        for (int j = 0; j < ndims; ++j) {
          double const dbasis_dx = dN[n][j];
          R_nodal[0][n][i] += stress(i, j) * dbasis_dx * w * thickness * dv;
        }
    return;
  }
  if (ip_set == 0) {  // evaluate_displacement, mechanics.cpp:116-145
    Tens<T> stress = local.cauchy(*this);
    if (local.is_finite_deformation()) stress = stress * cof_F;  // PK1 = sigma cof(F)
    for (int n = 0; n < nn; ++n)
      for (int i = 0; i < ndims; ++i)
        for (int j = 0; j < ndims; ++j) {
          double const dbasis_dx = dN[n][j];
          R_nodal[0][n][i] += stress(i, j) * dbasis_dx * w * dv;
        }
  }
  // evaluate_mixed, mechanics.cpp:148-227
  T const E = local.params[0];
  T const nu = local.params[1];
  T const mu = compute_mu(E, nu);
  T const p = scalar_x(1);
  T pressure_scale_factor = local.pressure_scale_factor();
  if (ip_set == 0) {
    Vec<T> const grad_p = grad_scalar_x(1);
    Tens<T> const I = eye<T>(ndims);
    T hydro_cauchy = local.hydro_cauchy(*this);
    for (int n = 0; n < nn; ++n) {
      double const basis = N[n];
      R_nodal[1][n][0] -= hydro_cauchy / pressure_scale_factor * basis * w * dv;
    }
    T const tau = stab_mult * 0.5 * h * h / mu;
    Tens<T> stab_matrix = tau * I;
    if (local.is_finite_deformation()) stab_matrix = stab_matrix * (transpose(cof_F) * cof_F) / det_F;
    for (int n = 0; n < nn; ++n)
      for (int i = 0; i < ndims; ++i)
        for (int j = 0; j < ndims; ++j) {
          double const dbasis_dx = dN[n][i];
          R_nodal[1][n][0] -= stab_matrix(i, j) * grad_p(j) * dbasis_dx * w * dv;
        }
  } else {
    for (int n = 0; n < nn; ++n) {
      double const basis = N[n];
      R_nodal[1][n][0] -= p / pressure_scale_factor * basis * w * dv;
    }
  }
}

template <class T> Local<T>* make_local(std::string const& type, int ndims = 3) {  // local_residual.cpp:893-933
  if (ndims == 2) {  // the 2-D decks of the reference that run the 3-D classes on 2 x 2 tensors
    if (type == "small_J2") return new SmallJ2<T>(2);
    if (type == "small_hill_plane_strain") return new SmallHillPlaneStrain<T>();
    if (type == "hyper_J2_plane_strain") return new HyperJ2PlaneStrain<T>();
    if (type == "hypo_hill_plane_strain") return new HypoHillPlaneStrain<T>();
    if (type == "small_hill_plane_stress") return new SmallHillPlaneStress<T>();  // these three pair with mechanics_plane_stress
    if (type == "hyper_J2_plane_stress") return new HyperJ2PlaneStress<T>();
    if (type == "hypo_hill_plane_stress") return new HypoHillPlaneStress<T>();
    return nullptr;
  }
  if (type == "elastic") return new Elastic<T>();
  if (type == "small_J2") return new SmallJ2<T>();
  if (type == "hyper_J2") return new HyperJ2<T>();
  if (type == "small_hill") return new SmallHill<T>();
  if (type == "isotropic_elastic") return new IsotropicElastic<T>();
  if (type == "hypo_hill") return new HypoHill<T>();
  if (type == "small_hosford") return new SmallHosford<T>();
  if (type == "hypo_hosford") return new HypoHosford<T>();
  if (type == "hypo_barlat") return new HypoBarlat<T>();
  return nullptr;
}

// ---------------------------------------------------------------------------
// Discretisation tables (disc.cpp:263-265 get_dof, :356-387 ghost graph,
// :414-459 scatter offsets, :461-484 elem lids) for a single part.
// ---------------------------------------------------------------------------
struct Ctx {
  ElemKit kit;
  int ndims = 3;
  int nnodes = 0, nelems = 0, nsets = 1;
  std::vector<double> coords;
  std::vector<int> conn;
  std::vector<int> extra_pairs;  // extra (row node, col node) graph entries (multi-part union pattern)
  std::vector<std::vector<int>> set_elems;
  std::vector<int> elem_set_of;
  std::vector<std::vector<int>> colors;  // element colouring for the multi-thread baseline only
  // node graph (sorted neighbour lists) and the 2x2 dof-level CSR blocks
  std::vector<int64_t> nodeptr;
  std::vector<int> nodeadj;
  std::vector<int64_t> rowptr[2][2];
  std::vector<int> colidx[2][2];
  std::vector<int> offsets[2][2];  // scatter_offsets[i][j][e*stride + r*dofs_j + c]
  // model
  std::string local_type;
  double stab_mult = 1.;
  int max_iters = 0;
  double abs_tol = 0., rel_tol = 0.;
  int nparams = 0;
  std::vector<double> params;  // [set][param]
  std::vector<std::vector<int>> active;  // active parameter indices per element set
  Local<double>* local_d = nullptr;
  Local<Fad>* local_f = nullptr;
  int nloc = 0;
  int ngpts = 0;  // coupled points per element = points of the local-state field
  int nres = 2;           // global residuals: 2 `mechanics`, 1 `mechanics_plane_stress`
  double thickness = 1.;
  // QoI: 0 = average displacement (avg_disp.cpp), 1 = calibration (calibration.cpp, 3-D form)
  int qoi_kind = 0;
  struct Calib {
    double balance = 0., weights[3] = {1., 1., 1.}, area = 0., dt_over_T = 1.;
    int comp = 0;                            // reaction force component
    std::vector<int> nf;                     // [nelems] nodes of the element's face on the displacement side set (0 = none)
    std::vector<int> fnodes;                 // [nelems][4] local node ids of that face (m_mapping_disp -> downward face)
    std::vector<unsigned> load_mask;         // [nelems] bit n set: local node n lies on the load plane (m_mapping_load)
    std::vector<double> u_meas;              // measured displacement of the current step, nodal [nnodes][3]
    double load_meas = 0., total_load = 0., load_mismatch = 0.;
  } cal;
  ~Ctx() { delete local_d; delete local_f; }
};

static void build_graph(Ctx& c) {
  int const nn = c.kit.nn;
  std::vector<std::vector<int>> adj(c.nnodes);
  for (int e = 0; e < c.nelems; ++e)
    for (int a = 0; a < nn; ++a)
      for (int b = 0; b < nn; ++b) adj[c.conn[e * nn + a]].push_back(c.conn[e * nn + b]);
  for (size_t q = 0; q + 1 < c.extra_pairs.size(); q += 2) adj[c.extra_pairs[q]].push_back(c.extra_pairs[q + 1]);
  c.nodeptr.assign(c.nnodes + 1, 0);
  for (int n = 0; n < c.nnodes; ++n) {
    std::sort(adj[n].begin(), adj[n].end());
    adj[n].erase(std::unique(adj[n].begin(), adj[n].end()), adj[n].end());
    c.nodeptr[n + 1] = c.nodeptr[n] + (int64_t)adj[n].size();
  }
  c.nodeadj.resize(c.nodeptr[c.nnodes]);
  for (int n = 0; n < c.nnodes; ++n) std::copy(adj[n].begin(), adj[n].end(), c.nodeadj.begin() + c.nodeptr[n]);
  int const neq[2] = {c.ndims, 1};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      auto& rp = c.rowptr[i][j];
      auto& ci = c.colidx[i][j];
      rp.assign((size_t)c.nnodes * neq[i] + 1, 0);
      ci.clear();
      ci.reserve((size_t)c.nodeptr[c.nnodes] * neq[i] * neq[j]);
      for (int n = 0; n < c.nnodes; ++n)
        for (int ei = 0; ei < neq[i]; ++ei) {
          for (int64_t k = c.nodeptr[n]; k < c.nodeptr[n + 1]; ++k)
            for (int ej = 0; ej < neq[j]; ++ej) ci.push_back(c.nodeadj[k] * neq[j] + ej);
          rp[(size_t)n * neq[i] + ei + 1] = (int64_t)ci.size();
        }
      // scatter offsets by lower_bound in each row, as disc.cpp:414-459
      int const dofs_i = neq[i] * nn, dofs_j = neq[j] * nn, stride = dofs_i * dofs_j;
      auto& so = c.offsets[i][j];
      so.assign((size_t)c.nelems * stride, -1);
      for (int e = 0; e < c.nelems; ++e)
        for (int in = 0; in < nn; ++in)
          for (int ie = 0; ie < neq[i]; ++ie) {
            int const row = c.conn[e * nn + in] * neq[i] + ie;
            int const* rb = ci.data() + rp[row];
            int const* re = ci.data() + rp[row + 1];
            int const row_off = (in * neq[i] + ie) * dofs_j;
            for (int jn = 0; jn < nn; ++jn)
              for (int je = 0; je < neq[j]; ++je) {
                int const col = c.conn[e * nn + jn] * neq[j] + je;
                int const* it = std::lower_bound(rb, re, col);
                so[(size_t)e * stride + row_off + jn * neq[j] + je] = (int)(it - ci.data());
              }
          }
    }
}

struct Fields {
  double const* u; double const* p; double const* u_prev; double const* p_prev;
  double const* xi_prev; double* xi;
};
struct LinSys { double* A[2][2]; double* b[2]; };

static void elem_coords(Ctx const& c, int e, double X[][3]) {
  for (int n = 0; n < c.kit.nn; ++n)
    for (int d = 0; d < 3; ++d) X[n][d] = c.coords[(size_t)c.conn[e * c.kit.nn + n] * 3 + d];
}
static double elem_size(Ctx const& c, double const X[][3]) {  // mechanics.cpp:103-113
  double h = 0.;
  for (int k = 0; k < c.kit.nedges; ++k) {
    double l2 = 0.;
    for (int d = 0; d < 3; ++d) { double const t = X[c.kit.edges[k][1]][d] - X[c.kit.edges[k][0]][d]; l2 += t * t; }
    double const l = std::sqrt(l2);
    h += l * l;
  }
  return std::sqrt(h / c.kit.nedges);
}

// scatter_lhs (global_residual.cpp:556-586) and scatter_rhs (:463-479)
static void scatter_lhs(Ctx const& c, Global<Fad> const& g, int e, double const* dtotal, int ld, LinSys& ls) {
  int const nn = c.kit.nn;
  for (int i = 0; i < g.nres; ++i)
    for (int j = 0; j < g.nres; ++j) {
      int const dofs_i = g.neq[i] * nn, dofs_j = g.neq[j] * nn;
      double* vals = ls.A[i][j];
      int const* offsets = c.offsets[i][j].data() + (size_t)e * dofs_i * dofs_j;
      for (int in = 0; in < nn; ++in)
        for (int ie = 0; ie < g.neq[i]; ++ie) {
          int const i_idx = g.dx_idx(i, in, ie);
          int const row_offset = (in * g.neq[i] + ie) * dofs_j;
          for (int jn = 0; jn < nn; ++jn)
            for (int je = 0; je < g.neq[j]; ++je) {
              int const j_idx = g.dx_idx(j, jn, je);
              vals[offsets[row_offset + jn * g.neq[j] + je]] += dtotal[i_idx * ld + j_idx];
            }
        }
    }
}
template <class T>
static void scatter_rhs(Ctx const& c, Global<T> const& g, int e, double const* rhs, LinSys& ls) {
  int const nn = c.kit.nn;
  for (int i = 0; i < g.nres; ++i)
    for (int n = 0; n < nn; ++n)
      for (int eq = 0; eq < g.neq[i]; ++eq)
        ls.b[i][c.conn[e * nn + n] * g.neq[i] + eq] += rhs[g.dx_idx(i, n, eq)];
}

// ---------------------------------------------------------------------------
// eval_forward_jacobian, evaluations.cpp:12-154.  Element range [e0,e1) within
// the set loop lets the multi-thread baseline give each thread a partition.
// ---------------------------------------------------------------------------
static int forward_jacobian(Ctx& c, Local<Fad>& local, Fields const& f, LinSys& ls, int set_filter,
                            int e_begin, int e_end, std::vector<int> const* elist = nullptr) {
  Global<Fad> global;
  global.stab_mult = c.stab_mult;
  global.before_elems(c.kit.nn, c.ndims, c.nres);
  global.thickness = c.thickness;
  int const nn = c.kit.nn, nd = global.ndofs, nl = local.ndofs;
  int nderivs = -1;
  std::vector<double> dC_dxi(64), dC_dx(8 * NMAX), dxi_dx(8 * NMAX), dtotal(NMAX * NMAX), resid(NMAX);
  for (int es = 0; es < c.nsets; ++es) {
    if (set_filter >= 0 && es != set_filter) continue;
    local.before_elems(&c.params[(size_t)es * c.nparams], c.nparams);
    for (int e : (elist ? *elist : c.set_elems[es])) {
      if (e < e_begin || e >= e_end) continue;
      if (elist && c.nsets > 1 && c.elem_set_of[e] != es) continue;
      double X[8][3], N[8], dN[8][3];
      elem_coords(c, e, X);
      global.h = elem_size(c, X);
      global.gather(f.u, f.p, f.u_prev, f.p_prev, &c.conn[e * nn]);
      for (int ip_set = 0; ip_set < 2; ++ip_set) {
        int const npts = c.kit.npts[ip_set];
        for (int pt = 0; pt < npts; ++pt) {
          double const w = c.kit.wts[ip_set][pt];
          double const dv = shape_global(c.kit.type, nn, X, c.kit.pts[ip_set][pt], N, dN);
          global.set_weights(N, dN);
          if (ip_set == 0) {
            size_t const q = ((size_t)e * c.ngpts + pt) * nl;
            global.interpolate();
            local.gather(&f.xi[q], &f.xi_prev[q]);
            nderivs = local.seed_wrt_xi();
            int path = local.solve_nonlinear(global);
            if (path == -1) return path;
            local.scatter(&f.xi[q]);
            local.jacobian(nderivs, dC_dxi.data());
            local.unseed_wrt_xi();
            nderivs = global.seed_wrt_x();
            global.interpolate();
            local.evaluate(global);
            local.jacobian(nderivs, dC_dx.data());
            for (int k = 0; k < nl * nd; ++k) dC_dx[k] = -dC_dx[k];
            full_piv_lu_solve(nl, nd, dC_dxi.data(), dC_dx.data(), dxi_dx.data());
            local.seed_wrt_x(nd, dxi_dx.data());
          } else {
            nderivs = global.seed_wrt_x();
            global.interpolate();
          }
          global.zero_residual();
          global.evaluate(local, w, dv, ip_set);
          global.jacobian(nderivs, dtotal.data());
          global.residual_values(resid.data());
          scatter_lhs(c, global, e, dtotal.data(), nd, ls);
          scatter_rhs(c, global, e, resid.data(), ls);
          global.unseed_wrt_x();
        }
      }
    }
  }
  return 0;
}

// eval_global_residual, evaluations.cpp:156-259 (error-estimation branch omitted).
// Deviation noted in SURVEY.md section 10: the reference gathers local point 0 for every
// point; with several coupled points per element (hex8) this restatement
// gathers point `pt`.  Identical on tet4 (one coupled point).
static void global_residual(Ctx& c, Local<double>& local, Fields const& f, LinSys& ls) {
  Global<double> global;
  global.stab_mult = c.stab_mult;
  global.before_elems(c.kit.nn, c.ndims, c.nres);
  global.thickness = c.thickness;
  int const nn = c.kit.nn, nl = local.ndofs;
  std::vector<double> resid(NMAX);
  for (int es = 0; es < c.nsets; ++es) {
    local.before_elems(&c.params[(size_t)es * c.nparams], c.nparams);
    for (int e : c.set_elems[es]) {
      double X[8][3], N[8], dN[8][3];
      elem_coords(c, e, X);
      global.h = elem_size(c, X);
      global.gather(f.u, f.p, f.u_prev, f.p_prev, &c.conn[e * nn]);
      for (int ip_set = 0; ip_set < 2; ++ip_set)
        for (int pt = 0; pt < c.kit.npts[ip_set]; ++pt) {
          double const w = c.kit.wts[ip_set][pt];
          double const dv = shape_global(c.kit.type, nn, X, c.kit.pts[ip_set][pt], N, dN);
          global.set_weights(N, dN);
          if (ip_set == 0) {
            size_t const q = ((size_t)e * c.ngpts + pt) * nl;
            local.gather(&f.xi[q], &f.xi_prev[q]);
          }
          global.interpolate();
          global.zero_residual();
          global.evaluate(local, w, dv, ip_set);
          global.residual_values(resid.data());
          scatter_rhs(c, global, e, resid.data(), ls);
        }
    }
  }
}

// AvgDisp::evaluate, avg_disp.cpp:16-33
template <class T> static T avg_disp_point(Global<T> const& g, double w, double dv) {
  T value_pt = 0.;
  Vec<T> const u = g.vector_x(0);
  for (int i = 0; i < g.ndims; ++i) value_pt += u(i) * w * dv;
  value_pt /= g.ndims;
  return value_pt;
}
static double dxq(Fad const& v, int j) { return v.dx(j); }

// ---- Calibration QoI, 3-D form (calibration.cpp) ------------------------------------------------------------
// Element faces in local node ids (this restatement's own numbering of the downward faces; the reference only
// uses the face to find its nodes).
static int const TET_FACES[4][3] = {{0, 1, 2}, {0, 1, 3}, {1, 2, 3}, {0, 2, 3}};
static int const HEX_FACES[6][4] = {{0, 1, 2, 3}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}, {4, 5, 6, 7}};

// face quadrature of order 2 (calibration.cpp:262-266): tri3 3-point rule, quad4 2x2 Gauss.  Returns the number of
// points; Nf = face shape functions, wdv = weight * getDV of the face at the point.
static int face_rule(int nf, double const X[][3], double Nf[][4], double* wdv) {
  auto cross_norm = [](double const a[3], double const b[3]) {
    double const c0 = a[1] * b[2] - a[2] * b[1], c1 = a[2] * b[0] - a[0] * b[2], c2 = a[0] * b[1] - a[1] * b[0];
    return std::sqrt(c0 * c0 + c1 * c1 + c2 * c2);
  };
  if (nf == 3) {
    double a[3], b[3];
    for (int d = 0; d < 3; ++d) { a[d] = X[1][d] - X[0][d]; b[d] = X[2][d] - X[0][d]; }
    double const dv = cross_norm(a, b);
    double const st[3][2] = {{1. / 6., 1. / 6.}, {2. / 3., 1. / 6.}, {1. / 6., 2. / 3.}};
    for (int q = 0; q < 3; ++q) {
      Nf[q][0] = 1. - st[q][0] - st[q][1]; Nf[q][1] = st[q][0]; Nf[q][2] = st[q][1]; Nf[q][3] = 0.;
      wdv[q] = dv / 6.;
    }
    return 3;
  }
  double const gp = 1. / std::sqrt(3.);
  int q = 0;
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < 2; ++i, ++q) {
      double const s = (i ? gp : -gp), t = (j ? gp : -gp);
      double const sn[4] = {-1., 1., 1., -1.}, tn[4] = {-1., -1., 1., 1.};
      double ds[3] = {0., 0., 0.}, dt[3] = {0., 0., 0.};
      for (int k = 0; k < 4; ++k) {
        Nf[q][k] = 0.25 * (1. + sn[k] * s) * (1. + tn[k] * t);
        for (int d = 0; d < 3; ++d) {
          ds[d] += 0.25 * sn[k] * (1. + tn[k] * t) * X[k][d];
          dt[d] += 0.25 * tn[k] * (1. + sn[k] * s) * X[k][d];
        }
      }
      wdv[q] = cross_norm(ds, dt);
    }
  return 4;
}
// area of a face by the one-point rule the reference uses (calibration.cpp:122-126)
static double face_area(int nf, double const X[][3]) {
  double Nf[4][4], wdv[4];
  if (nf == 3) { face_rule(3, X, Nf, wdv); return 3. * wdv[0]; }
  double ds[3] = {0., 0., 0.}, dt[3] = {0., 0., 0.};
  double const sn[4] = {-1., 1., 1., -1.}, tn[4] = {-1., -1., 1., 1.};
  for (int k = 0; k < 4; ++k) for (int d = 0; d < 3; ++d) { ds[d] += 0.25 * sn[k] * X[k][d]; dt[d] += 0.25 * tn[k] * X[k][d]; }
  double const c0 = ds[1] * dt[2] - ds[2] * dt[1], c1 = ds[2] * dt[0] - ds[0] * dt[2], c2 = ds[0] * dt[1] - ds[1] * dt[0];
  return 4. * std::sqrt(c0 * c0 + c1 * c1 + c2 * c2);
}

// compute_surface_mismatch (calibration.cpp:225-300).  At a face point only the face's nodes have non-zero shape
// functions, and there they equal the face's own shape functions, so global->interpolate(boundaryToElementXi(..))
// is the face interpolation of the nodal values.
template <class T> static T calib_surface(Ctx const& c, Global<T> const& g, int e) {
  auto const& cal = c.cal;
  int const nf = cal.nf[e], nn = c.kit.nn;
  int const* fn = &cal.fnodes[(size_t)e * 4];
  double X[4][3], Nf[4][4], wdv[4];
  for (int k = 0; k < nf; ++k)
    for (int d = 0; d < 3; ++d) X[k][d] = c.coords[(size_t)c.conn[e * nn + fn[k]] * 3 + d];
  int const nq = face_rule(nf, X, Nf, wdv);
  T mismatch = 0.;
  for (int q = 0; q < nq; ++q) {
    T qoi = 0.;
    for (int d = 0; d < c.ndims; ++d) {
      T u_fem = 0.;
      double u_meas = 0.;
      for (int k = 0; k < nf; ++k) {
        u_fem += g.x_nodal[0][fn[k]][d] * Nf[q][k];
        u_meas += cal.u_meas[(size_t)c.conn[e * nn + fn[k]] * c.ndims + d] * Nf[q][k];
      }
      qoi += cal.weights[d] * (u_fem - u_meas) * (u_fem - u_meas);
    }
    mismatch += 0.5 * qoi * wdv[q] / cal.area * cal.dt_over_T;
  }
  return mismatch;
}
// compute_load (calibration.cpp:302-343): the internal force of the coupled weak form at this point, summed
// over the element's nodes on the load plane
template <class T> static T calib_load(Ctx const& c, Global<T>& g, Local<T>& local, int e, double w, double dv) {
  g.zero_residual();
  g.evaluate(local, w, dv, 0);
  T load = 0.;
  for (int n = 0; n < c.kit.nn; ++n)
    if (c.cal.load_mask[e] & (1u << n)) load += g.R_nodal[0][n][c.cal.comp];
  g.zero_residual();
  return load;
}
// QoI<T>::evaluate at one coupled point: AvgDisp (avg_disp.cpp:16-33) or Calibration (calibration.cpp:418-480:
// the double instantiation has the displacement term only, the FADT one adds balance*dt/T*load_mismatch*load)
static double qoi_point(Ctx& c, Global<double>& g, Local<double>&, int e, double w, double dv) {
  if (c.qoi_kind == 0) return avg_disp_point(g, w, dv);
  return c.cal.nf[e] ? calib_surface(c, g, e) : 0.;
}
static Fad qoi_point(Ctx& c, Global<Fad>& g, Local<Fad>& local, int e, double w, double dv) {
  if (c.qoi_kind == 0) return avg_disp_point(g, w, dv);
  Fad v = 0.;
  if (c.cal.nf[e]) v += calib_surface(c, g, e);
  if (c.cal.load_mask[e]) v += c.cal.balance * c.cal.dt_over_T * c.cal.load_mismatch * calib_load(c, g, local, e, w, dv);
  return v;
}

// eval_adjoint_jacobian, evaluations.cpp:349-526 (QoI = average displacement)
static void adjoint_jacobian(Ctx& c, Local<Fad>& local, Fields const& f, double* g_hist, double const* f_hist,
                             LinSys& ls) {
  Global<Fad> global;
  global.stab_mult = c.stab_mult;
  global.before_elems(c.kit.nn, c.ndims, c.nres);
  global.thickness = c.thickness;
  int const nn = c.kit.nn, nd = global.ndofs, nl = local.ndofs;
  int nderivs = -1;
  std::vector<double> dC_dxi(64), dC_dx(8 * NMAX), dxi_dx(8 * NMAX), dtotal(NMAX * NMAX), dtotalT(NMAX * NMAX),
      rhs(NMAX), dJ_dx(NMAX), dJ_dxi(8);
  for (int es = 0; es < c.nsets; ++es) {
    local.before_elems(&c.params[(size_t)es * c.nparams], c.nparams);
    for (int e : c.set_elems[es]) {
      double X[8][3], N[8], dN[8][3];
      elem_coords(c, e, X);
      global.h = elem_size(c, X);
      global.gather(f.u, f.p, f.u_prev, f.p_prev, &c.conn[e * nn]);
      for (int ip_set = 0; ip_set < 2; ++ip_set)
        for (int pt = 0; pt < c.kit.npts[ip_set]; ++pt) {
          double const w = c.kit.wts[ip_set][pt];
          double const dv = shape_global(c.kit.type, nn, X, c.kit.pts[ip_set][pt], N, dN);
          global.set_weights(N, dN);
          if (ip_set == 0) {
            size_t const qp = (size_t)e * c.ngpts + pt;
            global.interpolate();
            local.gather(&f.xi[qp * nl], &f.xi_prev[qp * nl]);
            nderivs = local.seed_wrt_xi();
            local.evaluate(global);
            local.jacobian(nderivs, dC_dxi.data());
            local.unseed_wrt_xi();
            nderivs = global.seed_wrt_x();
            global.interpolate();
            local.evaluate(global);
            local.jacobian(nderivs, dC_dx.data());
            for (int k = 0; k < nl * nd; ++k) dC_dx[k] = -dC_dx[k];
            full_piv_lu_solve(nl, nd, dC_dxi.data(), dC_dx.data(), dxi_dx.data());
            local.seed_wrt_x(nd, dxi_dx.data());
            global.zero_residual();
            global.evaluate(local, w, dv, ip_set);
            global.jacobian(nderivs, dtotal.data());
            for (int r = 0; r < nd; ++r) for (int s = 0; s < nd; ++s) dtotalT[r * nd + s] = dtotal[s * nd + r];
            scatter_lhs(c, global, e, dtotalT.data(), nd, ls);
            local.unseed_wrt_xi();
            // dJ/dx with x seeded, xi plain (:469-471)
            Fad J = qoi_point(c, global, local, e, w, dv);
            for (int j = 0; j < nd; ++j) dJ_dx[j] = dxq(J, j);
            global.unseed_wrt_x();
            // dJ/dxi with xi seeded (:474-478)
            nderivs = local.seed_wrt_xi();
            global.interpolate();
            J = qoi_point(c, global, local, e, w, dv);
            for (int j = 0; j < nl; ++j) dJ_dxi[j] = dxq(J, j);
            local.unseed_wrt_xi();
            double* g_pt = &g_hist[qp * nl];
            double const* f_pt = &f_hist[qp * nd];
            for (int k = 0; k < nl; ++k) g_pt[k] -= dJ_dxi[k];
            for (int j = 0; j < nd; ++j) {
              double s = -dJ_dx[j] + f_pt[j];
              for (int k = 0; k < nl; ++k) s += dxi_dx[k * nd + j] * g_pt[k];
              rhs[j] = s;
            }
            scatter_rhs(c, global, e, rhs.data(), ls);
          } else {
            nderivs = global.seed_wrt_x();
            global.interpolate();
            global.zero_residual();
            global.evaluate(local, w, dv, ip_set);
            global.jacobian(nderivs, dtotal.data());
            for (int r = 0; r < nd; ++r) for (int s = 0; s < nd; ++s) dtotalT[r * nd + s] = dtotal[s * nd + r];
            scatter_lhs(c, global, e, dtotalT.data(), nd, ls);
            // the reference leaves x seeded here; the next gather/seed resets it
            global.unseed_wrt_x();
          }
        }
    }
  }
}

// solve_adjoint_local, evaluations.cpp:528-659
static void solve_adjoint_local(Ctx& c, Local<Fad>& local, Fields const& f, double const* z_u, double const* z_p,
                                double* phi, double* g_hist, double* f_hist) {
  Global<Fad> global;
  global.stab_mult = c.stab_mult;
  global.before_elems(c.kit.nn, c.ndims, c.nres);
  global.thickness = c.thickness;
  int const nn = c.kit.nn, nd = global.ndofs, nl = local.ndofs;
  int nderivs = -1;
  std::vector<double> dC(8 * NMAX), dR(NMAX * NMAX), A(64), rhs(8), phi_pt(8), z(NMAX);
  for (int es = 0; es < c.nsets; ++es) {
    local.before_elems(&c.params[(size_t)es * c.nparams], c.nparams);
    for (int e : c.set_elems[es]) {
      double X[8][3], N[8], dN[8][3];
      elem_coords(c, e, X);
      global.h = elem_size(c, X);
      global.gather(f.u, f.p, f.u_prev, f.p_prev, &c.conn[e * nn]);
      for (int n = 0; n < nn; ++n) {  // gather_adjoint, global_residual.cpp:423-438
        int const node = c.conn[e * nn + n];
        for (int eq = 0; eq < c.ndims; ++eq) z[global.dx_idx(0, n, eq)] = z_u[node * c.ndims + eq];
        if (global.nres == 2) z[global.dx_idx(1, n, 0)] = z_p[node];
      }
      for (int pt = 0; pt < c.kit.npts[0]; ++pt) {
        double const w = c.kit.wts[0][pt];
        double const dv = shape_global(c.kit.type, nn, X, c.kit.pts[0][pt], N, dN);
        global.set_weights(N, dN);
        size_t const qp = (size_t)e * c.ngpts + pt;
        global.interpolate();
        local.gather(&f.xi[qp * nl], &f.xi_prev[qp * nl]);
        nderivs = local.seed_wrt_xi();
        global.zero_residual();
        global.evaluate(local, w, dv, 0);
        local.evaluate(global);
        local.jacobian(nderivs, dC.data());   // nl x nl
        global.jacobian(nderivs, dR.data());  // nd x nl
        for (int r = 0; r < nl; ++r) for (int s = 0; s < nl; ++s) A[r * nl + s] = dC[s * nl + r];
        double const* g_pt = &g_hist[qp * nl];
        for (int k = 0; k < nl; ++k) {
          double s = 0.;
          for (int j = 0; j < nd; ++j) s += dR[j * nl + k] * z[j];
          rhs[k] = g_pt[k] - s;
        }
        full_piv_lu_solve(nl, 1, A.data(), rhs.data(), phi_pt.data());
        for (int k = 0; k < nl; ++k) phi[qp * nl + k] = phi_pt[k];
        // global history: f = -(dC/dx_prev)^T phi (:628-633)
        local.unseed_wrt_xi();
        nderivs = global.seed_wrt_x_prev();
        global.interpolate();
        local.evaluate(global);
        local.jacobian(nderivs, dC.data());  // nl x nd
        for (int j = 0; j < nd; ++j) {
          double s = 0.;
          for (int k = 0; k < nl; ++k) s += dC[k * nd + j] * phi_pt[k];
          f_hist[qp * nd + j] = -s;
        }
        // local history: g = -(dC/dxi_prev)^T phi (:636-642)
        global.unseed_wrt_x_prev();
        global.interpolate();
        nderivs = local.seed_wrt_xi_prev();
        local.evaluate(global);
        local.jacobian(nderivs, dC.data());  // nl x nl
        for (int j = 0; j < nl; ++j) {
          double s = 0.;
          for (int k = 0; k < nl; ++k) s += dC[k * nl + j] * phi_pt[k];
          g_hist[qp * nl + j] = -s;
        }
        local.unseed_wrt_xi_prev();
      }
    }
  }
}

// eval_qoi, evaluations.cpp:662-756 (average displacement)
static double eval_qoi(Ctx& c, Local<double>& local, Fields const& f) {
  Global<double> global;
  global.before_elems(c.kit.nn, c.ndims, c.nres);
  global.thickness = c.thickness;
  int const nn = c.kit.nn;
  double J = 0.;
  for (int es = 0; es < c.nsets; ++es)
    for (int e : c.set_elems[es]) {
      double X[8][3], N[8], dN[8][3];
      elem_coords(c, e, X);
      global.gather(f.u, f.p, f.u_prev, f.p_prev, &c.conn[e * nn]);
      for (int pt = 0; pt < c.kit.npts[0]; ++pt) {
        double const w = c.kit.wts[0][pt];
        double const dv = shape_global(c.kit.type, nn, X, c.kit.pts[0][pt], N, dN);
        global.set_weights(N, dN);
        global.interpolate();
        J += qoi_point(c, global, local, e, w, dv);
      }
    }
  if (c.qoi_kind == 1)  // Calibration::postprocess (calibration.cpp:374-381), one rank
    J += 0.5 * c.cal.balance * c.cal.dt_over_T * c.cal.load_mismatch * c.cal.load_mismatch;
  return J;
}

// preprocess_qoi (evaluations.cpp:262-347) + Calibration::preprocess / preprocess_finalize (calibration.cpp:345-372,
// 398-416): the total reaction load of the step and its mismatch with the measured load
static double qoi_preprocess(Ctx& c, Local<double>& local, Fields const& f) {
  if (c.qoi_kind != 1) return 0.;
  Global<double> global;
  global.stab_mult = c.stab_mult;
  global.before_elems(c.kit.nn, c.ndims, c.nres);
  global.thickness = c.thickness;
  int const nn = c.kit.nn, nl = local.ndofs;
  double total = 0.;
  for (int es = 0; es < c.nsets; ++es) {
    local.before_elems(&c.params[(size_t)es * c.nparams], c.nparams);
    for (int e : c.set_elems[es]) {
      if (!c.cal.load_mask[e]) continue;
      double X[8][3], N[8], dN[8][3];
      elem_coords(c, e, X);
      global.h = elem_size(c, X);
      global.gather(f.u, f.p, f.u_prev, f.p_prev, &c.conn[e * nn]);
      for (int pt = 0; pt < c.kit.npts[0]; ++pt) {
        double const w = c.kit.wts[0][pt];
        double const dv = shape_global(c.kit.type, nn, X, c.kit.pts[0][pt], N, dN);
        global.set_weights(N, dN);
        global.interpolate();
        size_t const qp = (size_t)e * c.ngpts + pt;
        local.gather(&f.xi[qp * nl], &f.xi_prev[qp * nl]);
        total += calib_load(c, global, local, e, w, dv);
      }
    }
  }
  c.cal.total_load = total;
  c.cal.load_mismatch = total - c.cal.load_meas;
  return total;
}

// eval_qoi_gradient, evaluations.cpp:758-925 (no DFAD/NN parameters)
// grad_abs (may be null; a test-side diagnostic, not part of the reference): the sum of the MAGNITUDES of every product
// that enters grad -- the scale against which a re-ordered evaluation of the same sums can be judged.
static void qoi_gradient(Ctx& c, Local<Fad>& local, Fields const& f, double const* z_u, double const* z_p,
                         double const* phi, double* grad, double* grad_abs = nullptr) {
  Global<Fad> global;
  global.stab_mult = c.stab_mult;
  global.before_elems(c.kit.nn, c.ndims, c.nres);
  global.thickness = c.thickness;
  int const nn = c.kit.nn, nd = global.ndofs, nl = local.ndofs;
  std::vector<double> dC(8 * NMAX), dR(NMAX * NMAX), z(NMAX);
  int gofs = 0;
  for (int es = 0; es < c.nsets; ++es) {
    int const nact = (int)c.active[es].size();
    int const* act = c.active[es].data();
    std::vector<double> es_grad(nact, 0.), es_abs(nact, 0.);
    local.before_elems(&c.params[(size_t)es * c.nparams], c.nparams);
    for (int e : c.set_elems[es]) {
      double X[8][3], N[8], dN[8][3];
      elem_coords(c, e, X);
      global.h = elem_size(c, X);
      global.gather(f.u, f.p, f.u_prev, f.p_prev, &c.conn[e * nn]);
      for (int n = 0; n < nn; ++n) {
        int const node = c.conn[e * nn + n];
        for (int eq = 0; eq < c.ndims; ++eq) z[global.dx_idx(0, n, eq)] = z_u[node * c.ndims + eq];
        if (global.nres == 2) z[global.dx_idx(1, n, 0)] = z_p[node];
      }
      for (int ip_set = 0; ip_set < 2; ++ip_set)
        for (int pt = 0; pt < c.kit.npts[ip_set]; ++pt) {
          double const w = c.kit.wts[ip_set][pt];
          double const dv = shape_global(c.kit.type, nn, X, c.kit.pts[ip_set][pt], N, dN);
          global.set_weights(N, dN);
          global.interpolate();
          int const nderivs = local.seed_wrt_params(nact, act);
          if (ip_set == 0) {
            size_t const qp = (size_t)e * c.ngpts + pt;
            local.gather(&f.xi[qp * nl], &f.xi_prev[qp * nl]);
            local.evaluate(global);
            local.jacobian(nderivs, dC.data());  // nl x nact
            for (int a = 0; a < nact; ++a) {
              double s = 0.;
              for (int k = 0; k < nl; ++k) s += dC[k * nderivs + a] * phi[qp * nl + k];
              es_grad[a] += s;
              for (int k = 0; k < nl; ++k) es_abs[a] += std::fabs(dC[k * nderivs + a] * phi[qp * nl + k]);
            }
            Fad const J = qoi_point(c, global, local, e, w, dv);
            for (int a = 0; a < nact; ++a) { es_grad[a] += dxq(J, a); es_abs[a] += std::fabs(dxq(J, a)); }
          }
          global.zero_residual();
          global.evaluate(local, w, dv, ip_set);
          global.jacobian(nderivs, dR.data());  // nd x nact
          for (int a = 0; a < nact; ++a) {
            double s = 0.;
            for (int j = 0; j < nd; ++j) s += dR[j * nderivs + a] * z[j];
            es_grad[a] += s;
            for (int j = 0; j < nd; ++j) es_abs[a] += std::fabs(dR[j * nderivs + a] * z[j]);
          }
          local.unseed_wrt_params(nact, act);
          // R_nodal keeps parameter derivatives until zero_residual at the next point
        }
    }
    for (int a = 0; a < nact; ++a) grad[gofs + a] = es_grad[a];  // scatter_es_gradient (:859-867)
    if (grad_abs) for (int a = 0; a < nact; ++a) grad_abs[gofs + a] = es_abs[a];
    gofs += nact;
  }
}

}  // namespace c8o

// ---------------------------------------------------------------------------
// C interface (ctypes-friendly).  u is [nnodes*3], p is [nnodes]; local state
// arrays are [nelems][ncoupled_pts][nloc]; A blocks are CSR values over the
// graphs returned by c8o_graph().  All outputs are accumulated into (+=), as
// scatter_lhs/scatter_rhs do; the caller zeroes first (primal.cpp:98).
// ---------------------------------------------------------------------------
using namespace c8o;

extern "C" {

void* c8o_create(int elem_type, int nnodes, int nelems, double const* coords, int const* conn,
                 int const* elem_set, int nsets, char const* local_type, double stab_mult, int max_iters,
                 double abs_tol, double rel_tol, double const* params, int nparams, int nextra, int const* extra_pairs) {
  if (elem_type != TET4 && elem_type != HEX8 && elem_type != TRI3) return nullptr;
  Ctx* c = new Ctx();
  c->kit = make_kit(elem_type);
  c->ndims = kit_dims(elem_type);  // coords stay [nnodes][3] (z = 0 in 2-D); u is [nnodes][ndims]
  c->nnodes = nnodes;
  c->nelems = nelems;
  c->nsets = nsets;
  c->coords.assign(coords, coords + (size_t)nnodes * 3);
  c->conn.assign(conn, conn + (size_t)nelems * c->kit.nn);
  c->set_elems.resize(nsets);
  c->elem_set_of.resize(nelems);
  for (int e = 0; e < nelems; ++e) {
    c->elem_set_of[e] = elem_set ? elem_set[e] : 0;
    c->set_elems[c->elem_set_of[e]].push_back(e);
  }
  c->local_type = local_type;
  if (c->local_type.size() > 13 && c->local_type.compare(c->local_type.size() - 13, 13, "_plane_stress") == 0) {
    c->nres = 1;          // the decks pair these models with `mechanics_plane_stress`: u only,
    c->kit.npts[1] = 0;   // one ip set (mechanics_plane_stress.cpp:35-38)
  }
  c->stab_mult = stab_mult;
  c->max_iters = max_iters;
  c->abs_tol = abs_tol;
  c->rel_tol = rel_tol;
  c->local_d = make_local<double>(c->local_type, c->ndims);
  c->local_f = make_local<Fad>(c->local_type, c->ndims);
  if (!c->local_d || c->local_d->num_params() != nparams) { delete c; return nullptr; }
  c->local_d->max_iters = c->local_f->max_iters = max_iters;
  c->local_d->abs_tol = c->local_f->abs_tol = abs_tol;
  c->local_d->rel_tol = c->local_f->rel_tol = rel_tol;
  c->nparams = nparams;
  c->params.assign(params, params + (size_t)nsets * nparams);
  c->active.assign(nsets, std::vector<int>());
  c->active[0].push_back(0);  // default: E of element set 0 (small_J2.cpp:96-98)
  c->nloc = c->local_d->ndofs;
  c->ngpts = c->kit.npts[0];
  if (nextra > 0) c->extra_pairs.assign(extra_pairs, extra_pairs + (size_t)nextra * 2);
  build_graph(*c);
  return c;
}
void c8o_destroy(void* h) { delete (Ctx*)h; }
int c8o_nloc(void* h) { return ((Ctx*)h)->nloc; }
int c8o_ndims(void* h) { return ((Ctx*)h)->ndims; }
int c8o_nres(void* h) { return ((Ctx*)h)->nres; }
void c8o_set_thickness(void* h, double t) { ((Ctx*)h)->thickness = t; }
// the `line search:` sublist of a local residual (line_search.hpp:40-49; the Hosford / Barlat models use it)
void c8o_set_local_line_search(void* h, double c1, double bmin, double bmax, int max_evals) {
  Ctx* c = (Ctx*)h;
  LineSearchParams p;
  p.c1 = c1; p.backtrack_min = bmin; p.backtrack_max = bmax; p.max_evals = max_evals;
  c->local_d->ls = p;
  c->local_f->ls = p;
}
int c8o_npts(void* h) { return ((Ctx*)h)->ngpts; }
void c8o_set_params(void* h, double const* params) {
  Ctx* c = (Ctx*)h;
  c->params.assign(params, params + (size_t)c->nsets * c->nparams);
}
// Calibration QoI (calibration.cpp:13-50 parameters; :55-160 before_elems).  faces: [nfaces][npf] global node ids
// of the displacement side set; load plane: nodes with |x[coord_idx] - coord_value| < coord_tol (qoi.cpp:160-198).
void c8o_set_calibration(void* h, int nfaces, int npf, int const* faces, double const* weights, double balance,
                         int coord_idx, double coord_value, double coord_tol, int comp, double dt_over_T) {
  Ctx* c = (Ctx*)h;
  auto& cal = c->cal;
  c->qoi_kind = 1;
  cal.balance = balance;
  for (int d = 0; d < 3; ++d) cal.weights[d] = weights ? weights[d] : 1.;
  cal.comp = comp;
  cal.dt_over_T = dt_over_T;
  int const nn = c->kit.nn;
  std::set<std::vector<int>> side;
  for (int f = 0; f < nfaces; ++f) {
    std::vector<int> key(faces + (size_t)f * npf, faces + (size_t)(f + 1) * npf);
    std::sort(key.begin(), key.end());
    side.insert(key);
  }
  cal.nf.assign(c->nelems, 0);
  cal.fnodes.assign((size_t)c->nelems * 4, 0);
  cal.load_mask.assign(c->nelems, 0u);
  cal.area = 0.;
  if (nn == 3) {
    // 2-D branch (calibration.cpp:76-104): the displacement mismatch is integrated over the ELEMENTS -- all of them, or,
    // with a distance field and threshold, those the caller lists (faces = element ids, npf = 1) -- with the order-2
    // rule of the triangle (compute_disp_mismatch, :163-222); the area is the sum of the element areas (:88-91, :97-100)
    std::set<int> listed(faces, faces + (faces ? nfaces : 0));
    for (int e = 0; e < c->nelems; ++e) {
      if (nfaces > 0 && !listed.count(e)) continue;
      cal.nf[e] = 3;
      double X[4][3];
      for (int k = 0; k < 3; ++k) {
        cal.fnodes[(size_t)e * 4 + k] = k;
        for (int q = 0; q < 3; ++q) X[k][q] = c->coords[(size_t)c->conn[e * nn + k] * 3 + q];
      }
      cal.area += face_area(3, X);
    }
  }
  int const nfe = (nn == 3) ? 0 : ((nn == 4) ? 4 : 6), nfn = (nn == 4) ? 3 : 4;
  for (int e = 0; e < c->nelems; ++e) {
    for (int d = 0; d < nfe; ++d) {  // downward faces; a later match overwrites an earlier one (:107-131)
      int const* loc = (nn == 4) ? TET_FACES[d] : HEX_FACES[d];
      std::vector<int> key(nfn);
      for (int k = 0; k < nfn; ++k) key[k] = c->conn[e * nn + loc[k]];
      std::sort(key.begin(), key.end());
      if (side.count(key)) {
        cal.nf[e] = nfn;
        double X[4][3];
        for (int k = 0; k < nfn; ++k) {
          cal.fnodes[(size_t)e * 4 + k] = loc[k];
          for (int q = 0; q < 3; ++q) X[k][q] = c->coords[(size_t)c->conn[e * nn + loc[k]] * 3 + q];
        }
        cal.area += face_area(nfn, X);  // every match adds its area, as the reference does
      }
    }
    for (int n = 0; n < nn; ++n)
      if (std::abs(c->coords[(size_t)c->conn[e * nn + n] * 3 + coord_idx] - coord_value) < coord_tol) cal.load_mask[e] |= 1u << n;
  }
  cal.u_meas.assign((size_t)c->nnodes * c->ndims, 0.);
}
void c8o_set_avg_disp(void* h) { ((Ctx*)h)->qoi_kind = 0; }
// measured data of the current step: nodal displacements ("measured_<step>" field) and the load (load input file)
void c8o_set_measured(void* h, double const* u_meas, double load_meas) {
  Ctx* c = (Ctx*)h;
  c->cal.u_meas.assign(u_meas, u_meas + (size_t)c->nnodes * c->ndims);
  c->cal.load_meas = load_meas;
}
// preprocess_qoi for the current step: returns the total reaction load; out = {area, total load, load mismatch}
double c8o_qoi_preprocess(void* h, double const* u, double const* p, double const* u_prev, double const* p_prev,
                          double const* xi_prev, double const* xi, double* out) {
  Ctx* c = (Ctx*)h;
  Fields f{u, p, u_prev, p_prev, xi_prev, const_cast<double*>(xi)};
  double const t = qoi_preprocess(*c, *c->local_d, f);
  if (out) { out[0] = c->cal.area; out[1] = c->cal.total_load; out[2] = c->cal.load_mismatch; }
  return t;
}
void c8o_set_active(void* h, int es, int nactive, int const* idx) {
  Ctx* c = (Ctx*)h;
  c->active[es].assign(idx, idx + nactive);
}
void c8o_init_variables(void* h, double* xi) {  // local_residual.cpp:35-74
  Ctx* c = (Ctx*)h;
  for (size_t q = 0; q < (size_t)c->nelems * c->ngpts; ++q) c->local_d->init_variables(&xi[q * c->nloc]);
}
int64_t c8o_graph_nnz(void* h, int i, int j) { return (int64_t)((Ctx*)h)->colidx[i][j].size(); }
void c8o_graph(void* h, int i, int j, int64_t* rowptr, int* colidx) {
  Ctx* c = (Ctx*)h;
  std::copy(c->rowptr[i][j].begin(), c->rowptr[i][j].end(), rowptr);
  std::copy(c->colidx[i][j].begin(), c->colidx[i][j].end(), colidx);
}

int c8o_forward_jacobian(void* h, double const* u, double const* p, double const* u_prev, double const* p_prev,
                         double const* xi_prev, double* xi, double* A00, double* A01, double* A10, double* A11,
                         double* b0, double* b1) {
  Ctx* c = (Ctx*)h;
  Fields f{u, p, u_prev, p_prev, xi_prev, xi};
  LinSys ls{{{A00, A01}, {A10, A11}}, {b0, b1}};
  return forward_jacobian(*c, *c->local_f, f, ls, -1, 0, c->nelems);
}

// Multi-thread CPU baseline: the analogue of one single-threaded MPI rank per core, which is how the
// reference scales (Kokkos Serial).  Elements are greedily coloured (same-colour elements share no node);
// within a colour every thread takes a contiguous slice and adds straight into the shared A and b, so no
// private copies of the matrix are needed.  The summation order differs from the serial path (colour by
// colour instead of element order); this entry point is used for timing only, never for parity.
static void color_elements(Ctx& c) {
  if (!c.colors.empty()) return;
  int const nn = c.kit.nn;
  std::vector<uint64_t> used(c.nnodes, 0);
  for (int e = 0; e < c.nelems; ++e) {
    uint64_t mask = 0;
    for (int a = 0; a < nn; ++a) mask |= used[c.conn[e * nn + a]];
    int col = 0;
    while ((mask >> col) & 1) ++col;
    for (int a = 0; a < nn; ++a) used[c.conn[e * nn + a]] |= (uint64_t)1 << col;
    if ((int)c.colors.size() <= col) c.colors.resize(col + 1);
    c.colors[col].push_back(e);
  }
}

int c8o_forward_jacobian_mt(void* h, int nthreads, double const* u, double const* p, double const* u_prev,
                            double const* p_prev, double const* xi_prev, double* xi, double* A00, double* A01,
                            double* A10, double* A11, double* b0, double* b1) {
  Ctx* c = (Ctx*)h;
  if (nthreads <= 1) return c8o_forward_jacobian(h, u, p, u_prev, p_prev, xi_prev, xi, A00, A01, A10, A11, b0, b1);
  color_elements(*c);
  std::vector<Local<Fad>*> locals(nthreads);
  for (int t = 0; t < nthreads; ++t) {
    locals[t] = make_local<Fad>(c->local_type);
    locals[t]->max_iters = c->max_iters; locals[t]->abs_tol = c->abs_tol; locals[t]->rel_tol = c->rel_tol;
  }
  int rc = 0;
  for (auto const& col : c->colors) {
    std::vector<int> status(nthreads, 0);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
      th.emplace_back([&, t]() {
        size_t const n = col.size();
        std::vector<int> mine(col.begin() + n * t / nthreads, col.begin() + n * (t + 1) / nthreads);
        Fields f{u, p, u_prev, p_prev, xi_prev, xi};
        LinSys ls{{{A00, A01}, {A10, A11}}, {b0, b1}};
        status[t] = forward_jacobian(*c, *locals[t], f, ls, -1, 0, c->nelems, &mine);
      });
    }
    for (auto& t : th) t.join();
    for (int t = 0; t < nthreads; ++t) if (status[t] != 0) rc = -1;
  }
  for (auto* l : locals) delete l;
  return rc;
}

void c8o_global_residual(void* h, double const* u, double const* p, double const* u_prev, double const* p_prev,
                         double const* xi_prev, double* xi, double* b0, double* b1) {
  Ctx* c = (Ctx*)h;
  Fields f{u, p, u_prev, p_prev, xi_prev, xi};
  LinSys ls{{{nullptr, nullptr}, {nullptr, nullptr}}, {b0, b1}};
  global_residual(*c, *c->local_d, f, ls);
}

void c8o_adjoint_jacobian(void* h, double const* u, double const* p, double const* u_prev, double const* p_prev,
                          double const* xi_prev, double* xi, double* g_hist, double const* f_hist, double* A00,
                          double* A01, double* A10, double* A11, double* b0, double* b1) {
  Ctx* c = (Ctx*)h;
  Fields f{u, p, u_prev, p_prev, xi_prev, xi};
  LinSys ls{{{A00, A01}, {A10, A11}}, {b0, b1}};
  adjoint_jacobian(*c, *c->local_f, f, g_hist, f_hist, ls);
}

void c8o_solve_adjoint_local(void* h, double const* u, double const* p, double const* u_prev, double const* p_prev,
                             double const* xi_prev, double* xi, double const* z_u, double const* z_p, double* phi,
                             double* g_hist, double* f_hist) {
  Ctx* c = (Ctx*)h;
  Fields f{u, p, u_prev, p_prev, xi_prev, xi};
  solve_adjoint_local(*c, *c->local_f, f, z_u, z_p, phi, g_hist, f_hist);
}

double c8o_eval_qoi(void* h, double const* u, double const* p) {
  Ctx* c = (Ctx*)h;
  Fields f{u, p, u, p, nullptr, nullptr};
  return eval_qoi(*c, *c->local_d, f);
}

void c8o_qoi_gradient(void* h, double const* u, double const* p, double const* u_prev, double const* p_prev,
                      double const* xi_prev, double* xi, double const* z_u, double const* z_p, double const* phi,
                      double* grad) {
  Ctx* c = (Ctx*)h;
  Fields f{u, p, u_prev, p_prev, xi_prev, xi};
  qoi_gradient(*c, *c->local_f, f, z_u, z_p, phi, grad);
}
// the same, also returning the sum of the magnitudes of all products summed into each component (test diagnostic)
void c8o_qoi_gradient_abs(void* h, double const* u, double const* p, double const* u_prev, double const* p_prev,
                          double const* xi_prev, double* xi, double const* z_u, double const* z_p, double const* phi,
                          double* grad, double* grad_abs) {
  Ctx* c = (Ctx*)h;
  Fields f{u, p, u_prev, p_prev, xi_prev, xi};
  qoi_gradient(*c, *c->local_f, f, z_u, z_p, phi, grad, grad_abs);
}

// element-level probes used by unit tests: shape functions and quadrature
int c8o_kit_npts(int elem_type, int ip_set) { return make_kit(elem_type).npts[ip_set]; }
void c8o_kit_point(int elem_type, int ip_set, int pt, double* xi3, double* w) {
  ElemKit k = make_kit(elem_type);
  for (int d = 0; d < 3; ++d) xi3[d] = k.pts[ip_set][pt][d];
  *w = k.wts[ip_set][pt];
}
double c8o_shape(int elem_type, double const* X, double const* xi3, double* N, double* dN) {
  ElemKit k = make_kit(elem_type);
  double Xe[8][3], dNe[8][3];
  for (int n = 0; n < k.nn; ++n) for (int d = 0; d < 3; ++d) Xe[n][d] = X[n * 3 + d];
  double const dv = shape_global(elem_type, k.nn, Xe, xi3, N, dNe);
  for (int n = 0; n < k.nn; ++n) for (int d = 0; d < 3; ++d) dN[n * 3 + d] = dNe[n][d];
  return dv;
}

}  // extern "C"
