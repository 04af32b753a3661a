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
