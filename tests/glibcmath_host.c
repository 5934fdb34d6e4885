/* host build of criteria3d_amd/csrc/sf3d_glibcmath.inc (same text as the device compiles) for tests/test_glibcmath.py:
 * the routines next to the C library's own, and counters of the arguments on which the two differ */
#include <math.h>
#include <stddef.h>
#include <string.h>
#define SF3D_GL_FN static inline
#define SF3D_GL_TABLE static const
#include "sf3d_glibcmath.inc"

static int same(double a, double b) { return (a != a && b != b) || memcmp(&a, &b, 8) == 0; }

void gl_log(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = sf3d_gl_log(x[i]); }
void gl_log_libm(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = log(x[i]); }
void gl_pow(const double* x, const double* y, double* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = sf3d_gl_pow(x[i], y[i]); }
void gl_pow_libm(const double* x, const double* y, double* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = pow(x[i], y[i]); }
void gl_exp(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = sf3d_gl_exp(x[i]); }
void gl_exp_libm(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = exp(x[i]); }
void gl_cbrt(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = sf3d_gl_cbrt(x[i]); }
void gl_cbrt_libm(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = cbrt(x[i]); }

/* number of arguments on which routine and library differ (a nan equals a nan); first[0..1] receives the first such argument */
size_t gl_count_diff1(int which, const double* x, size_t n, double* first)
{
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        double a, b;
        if (which == 0) { a = sf3d_gl_log(x[i]); b = log(x[i]); }
        else if (which == 1) { a = sf3d_gl_exp(x[i]); b = exp(x[i]); }
        else { a = sf3d_gl_cbrt(x[i]); b = cbrt(x[i]); }
        if (!same(a, b)) { if (!bad) first[0] = x[i]; ++bad; }
    }
    return bad;
}
size_t gl_count_diff_pow(const double* x, const double* y, size_t n, double* first)
{
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i)
        if (!same(sf3d_gl_pow(x[i], y[i]), pow(x[i], y[i]))) { if (!bad) { first[0] = x[i]; first[1] = y[i]; } ++bad; }
    return bad;
}
