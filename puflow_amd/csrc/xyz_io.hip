// Host-side text formatting of the CLI's output clouds (the on-disk format after the path).  The reference
// writes np.savetxt(path, cloud, fmt='%.6f') (modules/discrete/upsample.py:57): one line per point, values separated by one
// blank, printf("%.6f") of the float32 value.  With the GPU part of a cloud at ~7 ms, Python's 13-36 ms of formatting was
// what the CLI waited for; this writes the same bytes at ~1 ms per 20 000-point cloud.
//
// Exactness: a float32 is m * 2^e with a 24-bit m, 10^6 = 2^6 * 15625 (14 bits), so the product value * 10^6 has at most 38
// significant bits and is EXACT in a double.  "%.6f" is that product rounded to an integer, ties to even (glibc and CPython
// both format correctly rounded) = nearbyint() in the default rounding mode.  Non-finite values and magnitudes beyond 2^52
// go through snprintf.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "pf_api_internal.h"

namespace {

inline char* put_u64(char* p, unsigned long long v) {       // decimal digits of v, no padding
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

inline char* put_f6(char* p, float f) {
    const double v = (double)f;
    if (std::isnan(v)) { memcpy(p, "nan", 3); return p + 3; }                 // Python prints no sign for a NaN
    if (!std::isfinite(v) || std::fabs(v) >= 4.0e9) return p + snprintf(p, 64, "%.6f", v);
    const double r = std::nearbyint(std::fabs(v) * 1e6);    // exact product, ties to even
    const unsigned long long u = (unsigned long long)r;
    if (std::signbit(v)) *p++ = '-';                        // printf keeps the sign of values that round to zero
    p = put_u64(p, u / 1000000ull);
    *p++ = '.';
    unsigned frac = (unsigned)(u % 1000000ull);
    for (int i = 5; i >= 0; --i) { p[i] = (char)('0' + frac % 10); frac /= 10; }
    return p + 6;
}

}  // namespace

// Upper bound of the bytes pf_format_xyz writes for n rows of c values
extern "C" long long pf_format_xyz_bound(long long n, int c) {
    if (n < 0 || c <= 0) return -1;
    return n * ((long long)c * 64 + 1) + 1;
}

// pts [n,c] float32 (host memory) -> text in out (capacity cap bytes): rows of c "%.6f" values separated by ' ', '\n' after
// every row - the bytes np.savetxt(fmt='%.6f') writes.  Returns the number of bytes written, or a negative PF_ERR_* code.
extern "C" long long pf_format_xyz(const float* pts, long long n, int c, char* out, long long cap) {
    if (!pts || !out) return PF_ERR_NULL;
    if (n < 0 || c <= 0) return PF_ERR_SHAPE;
    if (cap < pf_format_xyz_bound(n, c)) return PF_ERR_WORKSPACE;
    char* p = out;
    for (long long i = 0; i < n; ++i) {
        for (int j = 0; j < c; ++j) {
            if (j) *p++ = ' ';
            p = put_f6(p, pts[i * c + j]);
        }
        *p++ = '\n';
    }
    return (long long)(p - out);
}

// Reader for the CLI's input clouds: the values np.loadtxt(path, dtype=np.float32) returns (modules/discrete/upsample.py:42)
// for whitespace-separated numeric text - every token parsed as a double (strtod, "C" locale) and rounded to float32, '#'
// starts a comment, blank lines are skipped, every data row must have the same number of columns.  text[0..len) must be
// followed by a terminating 0 byte (text[len] == 0).  Returns the number of values written to out (rows x *ncols), or
// PF_ERR_WORKSPACE when out is too small, PF_ERR_SHAPE for ragged rows, PF_ERR_UNSUPPORTED for a token that is not a number
// (the caller then falls back to numpy, which knows more dialects).
extern "C" long long pf_parse_xyz(const char* text, long long len, float* out, long long cap, int* ncols) {
    if (!text || !out || !ncols) return PF_ERR_NULL;
    if (len < 0 || text[len] != 0) return PF_ERR_SHAPE;
    const char* p = text;
    const char* end = text + len;
    long long n = 0;
    int cols = -1;
    while (p < end) {
        const char* eol = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
        if (!eol) eol = end;
        const char* stop = static_cast<const char*>(memchr(p, '#', (size_t)(eol - p)));
        if (!stop) stop = eol;
        int c = 0;
        while (p < stop) {
            while (p < stop && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\v' || *p == '\f')) ++p;
            if (p >= stop) break;
            char* q = nullptr;
            double v;
            {
                // Clinger's fast path for plain decimals ([-+]digits[.digits], at most 15 significant digits): the digits
                // as an integer < 2^53 and a power of ten <= 10^22 are both exact doubles, so one division is the correctly
                // rounded value - the same double strtod returns.  Everything else (exponents, long digit strings, nan / inf,
                // hex floats) goes to strtod.
                static const double P10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
                const char* t = p;
                const bool neg = *t == '-';
                if (*t == '-' || *t == '+') ++t;
                unsigned long long m = 0;
                int nd = 0, frac = 0;
                while (t < stop && *t >= '0' && *t <= '9' && nd < 16) { m = m * 10 + (unsigned)(*t - '0'); ++t; ++nd; }
                if (t < stop && *t == '.') {
                    ++t;
                    while (t < stop && *t >= '0' && *t <= '9' && nd < 16) { m = m * 10 + (unsigned)(*t - '0'); ++t; ++nd; ++frac; }
                }
                const bool ends = t >= stop || *t == ' ' || *t == '\t' || *t == '\r' || *t == '\v' || *t == '\f';
                if (ends && nd >= 1 && nd <= 15) {
                    v = (double)m / P10[frac];
                    if (neg) v = -v;
                    q = const_cast<char*>(t);
                } else {
                    v = strtod(p, &q);                      // never runs past the terminating 0
                }
            }
            if (q == p || q > stop) return PF_ERR_UNSUPPORTED;
            if (q < stop && !(*q == ' ' || *q == '\t' || *q == '\r' || *q == '\v' || *q == '\f')) return PF_ERR_UNSUPPORTED;
            if (n >= cap) return PF_ERR_WORKSPACE;
            out[n++] = (float)v;
            ++c;
            p = q;
        }
        if (c > 0) {
            if (cols < 0) cols = c;
            else if (c != cols) return PF_ERR_SHAPE;
        }
        p = eol < end ? eol + 1 : end;
    }
    *ncols = cols < 0 ? 0 : cols;
    return n;
}
