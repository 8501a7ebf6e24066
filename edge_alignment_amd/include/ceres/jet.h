// ceres/jet.h — the slice of ceres::Jet<T, N> that the reference's cost functors touch when they are instantiated for
// automatic differentiation (standalone/utils.h:47-80, include/EAResidue.h:85-118 are `template <typename T>`): a value
// and N partial derivatives with +, -, *, / between jets and scalars, unary minus and comparisons on the value.
// The drop-in's solver never differentiates on the host (the 1x6 row is analytic, in the gfx950 kernels); this type exists
// so that the templated operator() of the functors stays instantiable for host-side spot checks, as upstream.
#pragma once

namespace ceres {

template <typename T, int N>
struct Jet {
  enum { DIMENSION = N };
  T a;
  T v[N];
  Jet() : a(T()) { for (int i = 0; i < N; ++i) v[i] = T(); }
  Jet(const T &value) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(); }  // NOLINT: scalars promote, as in Ceres
  Jet(const T &value, int k) : a(value) { for (int i = 0; i < N; ++i) v[i] = (i == k) ? T(1) : T(); }
};

template <typename T, int N> Jet<T, N> operator+(const Jet<T, N> &f) { return f; }
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N> &f) {
  Jet<T, N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h;
}
template <typename T, int N> Jet<T, N> operator+(const Jet<T, N> &f, const Jet<T, N> &g) {
  Jet<T, N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h;
}
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N> &f, const Jet<T, N> &g) {
  Jet<T, N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h;
}
template <typename T, int N> Jet<T, N> operator*(const Jet<T, N> &f, const Jet<T, N> &g) {
  Jet<T, N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h;
}
template <typename T, int N> Jet<T, N> operator/(const Jet<T, N> &f, const Jet<T, N> &g) {
  // (f / g)' = (f' - (f / g) g') / g, the form ceres/jet.h uses
  Jet<T, N> h;
  const T gi = T(1) / g.a;
  h.a = f.a * gi;
  for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - h.a * g.v[i]) * gi;
  return h;
}
// jet (op) scalar, scalar (op) jet
template <typename T, int N> Jet<T, N> operator+(const Jet<T, N> &f, T s) { Jet<T, N> h = f; h.a += s; return h; }
template <typename T, int N> Jet<T, N> operator+(T s, const Jet<T, N> &f) { Jet<T, N> h = f; h.a += s; return h; }
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N> &f, T s) { Jet<T, N> h = f; h.a -= s; return h; }
template <typename T, int N> Jet<T, N> operator-(T s, const Jet<T, N> &f) { Jet<T, N> h = -f; h.a += s; return h; }
template <typename T, int N> Jet<T, N> operator*(const Jet<T, N> &f, T s) {
  Jet<T, N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h;
}
template <typename T, int N> Jet<T, N> operator*(T s, const Jet<T, N> &f) { return f * s; }
template <typename T, int N> Jet<T, N> operator/(const Jet<T, N> &f, T s) { return f * (T(1) / s); }
template <typename T, int N> Jet<T, N> operator/(T s, const Jet<T, N> &g) { return Jet<T, N>(s) / g; }
template <typename T, int N> Jet<T, N> &operator+=(Jet<T, N> &f, const Jet<T, N> &g) { f = f + g; return f; }
template <typename T, int N> Jet<T, N> &operator-=(Jet<T, N> &f, const Jet<T, N> &g) { f = f - g; return f; }
template <typename T, int N> Jet<T, N> &operator*=(Jet<T, N> &f, const Jet<T, N> &g) { f = f * g; return f; }
template <typename T, int N> Jet<T, N> &operator/=(Jet<T, N> &f, const Jet<T, N> &g) { f = f / g; return f; }
// comparisons look at the value only (ceres/jet.h)
#define EA_JET_CMP(op)                                                                                   \
  template <typename T, int N> bool operator op(const Jet<T, N> &f, const Jet<T, N> &g) { return f.a op g.a; } \
  template <typename T, int N> bool operator op(const Jet<T, N> &f, T s) { return f.a op s; }               \
  template <typename T, int N> bool operator op(T s, const Jet<T, N> &g) { return s op g.a; }
EA_JET_CMP(<) EA_JET_CMP(<=) EA_JET_CMP(>) EA_JET_CMP(>=) EA_JET_CMP(==) EA_JET_CMP(!=)
#undef EA_JET_CMP

}  // namespace ceres
