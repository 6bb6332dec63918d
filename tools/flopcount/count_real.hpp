// count_real.hpp -- an arithmetic type that counts floating-point operations, substituted for ORC_REAL when
// the oracle (oracle/irrl_oracle.c, valid C++) is compiled by tools/flopcount.  Convention:
// add / sub / mul / div / sqrt / transcendental = 1 flop each, comparisons and negation = 0.
#pragma once
#include <cmath>
#include <cstdint>

struct CountReal {
  double v;
  static thread_local uint64_t n;
  CountReal() : v(0) {}
  CountReal(double x) : v(x) {}
  CountReal(float x) : v(x) {}
  CountReal(int x) : v(x) {}
  explicit operator double() const { return v; }
  explicit operator float() const { return (float)v; }
};
inline CountReal operator+(CountReal a, CountReal b) { CountReal::n++; return CountReal(a.v + b.v); }
inline CountReal operator-(CountReal a, CountReal b) { CountReal::n++; return CountReal(a.v - b.v); }
inline CountReal operator*(CountReal a, CountReal b) { CountReal::n++; return CountReal(a.v * b.v); }
inline CountReal operator/(CountReal a, CountReal b) { CountReal::n++; return CountReal(a.v / b.v); }
inline CountReal operator-(CountReal a) { return CountReal(-a.v); }
inline CountReal &operator+=(CountReal &a, CountReal b) { a = a + b; return a; }
inline CountReal &operator-=(CountReal &a, CountReal b) { a = a - b; return a; }
inline CountReal &operator*=(CountReal &a, CountReal b) { a = a * b; return a; }
inline bool operator<(CountReal a, CountReal b) { return a.v < b.v; }
inline bool operator>(CountReal a, CountReal b) { return a.v > b.v; }
inline bool operator<=(CountReal a, CountReal b) { return a.v <= b.v; }
inline bool operator>=(CountReal a, CountReal b) { return a.v >= b.v; }
inline bool operator==(CountReal a, CountReal b) { return a.v == b.v; }
inline bool operator!=(CountReal a, CountReal b) { return a.v != b.v; }
inline bool operator!(CountReal a) { return !(a.v != 0.0); }

#define ORC_CUSTOM_MATH 1
#define CR1(name, fn) inline CountReal name(CountReal x) { CountReal::n++; return CountReal(fn(x.v)); }
CR1(R_SQRT, std::sqrt) CR1(R_SIN, std::sin) CR1(R_COS, std::cos) CR1(R_ASIN, std::asin) CR1(R_ACOS, std::acos)
CR1(R_EXP, std::exp) CR1(R_LOG, std::log)
inline CountReal R_FABS(CountReal x) { return CountReal(std::fabs(x.v)); }
inline CountReal R_FLOOR(CountReal x) { CountReal::n++; return CountReal(std::floor(x.v)); }
inline CountReal R_FMOD(CountReal x, CountReal y) { CountReal::n++; return CountReal(std::fmod(x.v, y.v)); }
inline double R_TO_DOUBLE(CountReal x) { return x.v; }
