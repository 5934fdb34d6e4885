/* host build of criteria3d_amd/csrc/sf3d_fastmath.inc (same text as the device compiles) for tests/test_fastmath.py */
#include <math.h>
#include <stddef.h>
#define SF3D_FM_FN static inline
#define SF3D_FM_TABLE static const
#include "sf3d_fastmath.inc"

void fm_log(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = sf3d_flog(x[i]); }
void fm_log_libm(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = log(x[i]); }
void fm_pow(const double* x, const double* y, double* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = sf3d_fpow(x[i], y[i]); }
void fm_pow_libm(const double* x, const double* y, double* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = pow(x[i], y[i]); }
void fm_exp(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = sf3d_fexp(x[i]); }
void fm_exp_libm(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = exp(x[i]); }
void fm_cbrt(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = sf3d_fcbrt(x[i]); }
void fm_cbrt_libm(const double* x, double* y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = cbrt(x[i]); }
