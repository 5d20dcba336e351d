// TEST INFRASTRUCTURE -- a counting scalar for the INSTRUMENTED build of the CPU oracle (liborc_flops.so, -DORC_COUNT_FLOPS).
//
// SURVEY.md section 8d asks for the algorithmic fp64 flop count of the eye pass "counted exactly (instrumented build)" rather
// than estimated: with this header the oracle's `real` is a struct that wraps a double and counts every arithmetic operation
// the restated algorithm performs -- additions / subtractions, multiplications, divisions, square roots, and calls of
// pow / sin / cos / atan / exp (transcendental: listed, not folded into the flop total).  Comparisons, negations, fabs, floor
// and conversions are not counted.  Values are unchanged (every operation is the same IEEE operation on the wrapped double),
// so the instrumented build renders the same image as liborc.so -- tests/test_oracle_flops.py checks that.
// Counters are per thread and summed by orc_flop_counts().
#ifndef CGRT_FLOPCOUNT_H
#define CGRT_FLOPCOUNT_H
#include <cmath>
#include <cstdint>
#include <type_traits>

namespace orc {

enum { FC_ADD = 0, FC_MUL = 1, FC_DIV = 2, FC_SQRT = 3, FC_TRANS = 4, FC_N = 8 };
struct FlopCounters {
    uint64_t c[FC_N];
};
extern thread_local FlopCounters g_fc;

struct real {
    double v;
    real() : v(0) {}
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
    real(T x) : v((double)x) {}
    operator double() const { return v; }  // implicit: results flow out through the C API as plain doubles
    explicit operator float() const { return (float)v; }
    explicit operator int() const { return (int)v; }
    explicit operator long() const { return (long)v; }
    explicit operator long long() const { return (long long)v; }
    explicit operator unsigned long() const { return (unsigned long)v; }
    explicit operator bool() const { return v != 0; }
    real operator-() const { real r; r.v = -v; return r; }
    real &operator+=(real b) { g_fc.c[FC_ADD]++; v += b.v; return *this; }
    real &operator-=(real b) { g_fc.c[FC_ADD]++; v -= b.v; return *this; }
    real &operator*=(real b) { g_fc.c[FC_MUL]++; v *= b.v; return *this; }
    real &operator/=(real b) { g_fc.c[FC_DIV]++; v /= b.v; return *this; }
};
#define ORC_ARITH(T) typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0
#define ORC_BINOP(op, slot)                                                                                            \
    inline real operator op(real a, real b) { g_fc.c[slot]++; real r; r.v = a.v op b.v; return r; }                     \
    template <class T, ORC_ARITH(T)> inline real operator op(real a, T b) { g_fc.c[slot]++; real r; r.v = a.v op (double)b; return r; } \
    template <class T, ORC_ARITH(T)> inline real operator op(T a, real b) { g_fc.c[slot]++; real r; r.v = (double)a op b.v; return r; }
ORC_BINOP(+, FC_ADD)
ORC_BINOP(-, FC_ADD)
ORC_BINOP(*, FC_MUL)
ORC_BINOP(/, FC_DIV)
#undef ORC_BINOP
#define ORC_CMP(op)                                                                                   \
    inline bool operator op(real a, real b) { return a.v op b.v; }                                     \
    template <class T, ORC_ARITH(T)> inline bool operator op(real a, T b) { return a.v op (double)b; } \
    template <class T, ORC_ARITH(T)> inline bool operator op(T a, real b) { return (double)a op b.v; }
ORC_CMP(<)
ORC_CMP(>)
ORC_CMP(<=)
ORC_CMP(>=)
ORC_CMP(==)
ORC_CMP(!=)
#undef ORC_CMP
inline real sqrt(real a) { g_fc.c[FC_SQRT]++; return real(std::sqrt(a.v)); }
inline real fabs(real a) { return real(std::fabs(a.v)); }
inline real floor(real a) { return real(std::floor(a.v)); }
inline real ceil(real a) { return real(std::ceil(a.v)); }
inline real sin(real a) { g_fc.c[FC_TRANS]++; return real(std::sin(a.v)); }
inline real cos(real a) { g_fc.c[FC_TRANS]++; return real(std::cos(a.v)); }
inline real atan(real a) { g_fc.c[FC_TRANS]++; return real(std::atan(a.v)); }
inline real exp(real a) { g_fc.c[FC_TRANS]++; return real(std::exp(a.v)); }
template <class T> inline real pow(real a, T b) { g_fc.c[FC_TRANS]++; return real(std::pow(a.v, (double)b)); }
inline double plain(real a) { return a.v; }

}  // namespace orc
#endif
