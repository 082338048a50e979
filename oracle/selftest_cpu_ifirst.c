/*
 * ORACLE (test infrastructure, NOT product code): sanitizer self-test of oracle/cpu_ifirst.c.
 *
 * Sanitizers belong on the CPU build (GPU AddressSanitizer is not available on this pool).  This driver includes the C
 * restatement, runs every entry point on heap arrays allocated EXACTLY as large as the stencil's reach requires -- a read
 * or write one element outside is a heap-buffer-overflow under -fsanitize=address --, on I-contiguous and on strided
 * layouts and on degenerate domains, and checks a known answer per stencil.  tests/test_oracle.py builds it with
 * -fsanitize=address,undefined -fno-sanitize-recover=all and expects exit status 0.
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "cpu_ifirst.c"

static int failures = 0;
#define CHECK(cond, ...)                    \
    do {                                    \
        if (!(cond)) {                      \
            ++failures;                     \
            fprintf(stderr, __VA_ARGS__);   \
            fprintf(stderr, "\n");          \
        }                                   \
    } while (0)

/* element (i, j, k) of an (ni, nj, nk) array: layout 0 = I contiguous (gt:cpu_ifirst), 1 = K contiguous (numpy) */
static void strides_of(int layout, int64_t ni, int64_t nj, int64_t nk, int64_t* si, int64_t* sj, int64_t* sk) {
    if (layout == 0) { *si = 1; *sj = ni; *sk = ni * nj; }
    else { *sk = 1; *sj = nk; *si = nj * nk; }
}

static void test_lap(int64_t di, int64_t dj, int64_t dk, int layout) {
    const int64_t ni = di + 2, nj = dj + 2, nk = dk;
    int64_t si, sj, sk;
    strides_of(layout, ni, nj, nk, &si, &sj, &sk);
    double* in = malloc(sizeof(double) * (size_t)(ni * nj * nk + 1));   /* + 1: malloc(0) for empty domains */
    double* out = malloc(sizeof(double) * (size_t)(di * dj * dk + 1));
    int64_t osi, osj, osk;
    strides_of(layout, di, dj, dk, &osi, &osj, &osk);
    for (int64_t i = 0; i < ni; ++i)
        for (int64_t j = 0; j < nj; ++j)
            for (int64_t k = 0; k < nk; ++k) in[i * si + j * sj + k * sk] = (double)(i * i + j * j);  /* lap(x^2 + y^2) == 4 */
    oracle_lap5_f64(in + si + sj, si, sj, sk, out, osi, osj, osk, di, dj, dk);
    for (int64_t n = 0; n < di * dj * dk; ++n) CHECK(out[n] == 4.0, "lap5 %lldx%lldx%lld layout %d: %g", (long long)di, (long long)dj, (long long)dk, layout, out[n]);
    free(in);
    free(out);
}

static void test_hdiff(int64_t di, int64_t dj, int64_t dk, int layout, int limiter) {
    const int64_t ni = di + 4, nj = dj + 4, nk = dk;
    int64_t si, sj, sk, osi, osj, osk;
    strides_of(layout, ni, nj, nk, &si, &sj, &sk);
    strides_of(layout, di, dj, dk, &osi, &osj, &osk);
    double* in = malloc(sizeof(double) * (size_t)(ni * nj * nk + 1));
    double* cf = malloc(sizeof(double) * (size_t)(di * dj * dk + 1));
    double* out = malloc(sizeof(double) * (size_t)(di * dj * dk + 1));
    float* inf_ = malloc(sizeof(float) * (size_t)(ni * nj * nk + 1));
    float* cff = malloc(sizeof(float) * (size_t)(di * dj * dk + 1));
    float* outf = malloc(sizeof(float) * (size_t)(di * dj * dk + 1));
    for (int64_t i = 0; i < ni; ++i)
        for (int64_t j = 0; j < nj; ++j)
            for (int64_t k = 0; k < nk; ++k) {  /* a plane: lap == 0, every flux 0, out == in */
                in[i * si + j * sj + k * sk] = 3.0 * (double)i + 5.0 * (double)j + 7.0;
                inf_[i * si + j * sj + k * sk] = (float)(3 * i + 5 * j + 7);
            }
    for (int64_t n = 0; n < di * dj * dk; ++n) { cf[n] = 0.25; cff[n] = 0.25f; }
    oracle_hdiff_f64(in + 2 * si + 2 * sj, si, sj, sk, out, osi, osj, osk, cf, osi, osj, osk, di, dj, dk, limiter);
    oracle_hdiff_f32(inf_ + 2 * si + 2 * sj, si, sj, sk, outf, osi, osj, osk, cff, osi, osj, osk, di, dj, dk, limiter);
    for (int64_t i = 0; i < di; ++i)
        for (int64_t j = 0; j < dj; ++j)
            for (int64_t k = 0; k < dk; ++k) {
                const double want = 3.0 * (double)(i + 2) + 5.0 * (double)(j + 2) + 7.0;
                CHECK(out[i * osi + j * osj + k * osk] == want, "hdiff f64 %lldx%lldx%lld layout %d", (long long)di, (long long)dj, (long long)dk, layout);
                CHECK(outf[i * osi + j * osj + k * osk] == (float)want, "hdiff f32 %lldx%lldx%lld layout %d", (long long)di, (long long)dj, (long long)dk, layout);
            }
    free(in); free(cf); free(out); free(inf_); free(cff); free(outf);
}

static void test_tridiag(int64_t di, int64_t dj, int64_t dk, int layout) {
    int64_t si, sj, sk;
    strides_of(layout, di, dj, dk, &si, &sj, &sk);
    const size_t n = (size_t)(di * dj * dk);
    double *a = malloc(sizeof(double) * (n + 1)), *b = malloc(sizeof(double) * (n + 1)), *c = malloc(sizeof(double) * (n + 1));
    double *d = malloc(sizeof(double) * (n + 1)), *x = malloc(sizeof(double) * (n + 1));
    double *c0 = malloc(sizeof(double) * (n + 1)), *d0 = malloc(sizeof(double) * (n + 1));
    unsigned s = 12345u;
    for (size_t m = 0; m < n; ++m) {
        s = s * 1664525u + 1013904223u; a[m] = (double)(s >> 8) / 16777216.0 * 2.0 - 1.0;
        s = s * 1664525u + 1013904223u; c[m] = (double)(s >> 8) / 16777216.0 * 2.0 - 1.0;
        s = s * 1664525u + 1013904223u; d[m] = (double)(s >> 8) / 16777216.0 * 20.0 - 10.0;
        b[m] = 4.5;
        c0[m] = c[m]; d0[m] = d[m];
    }
    oracle_tridiag_f64(a, b, c, d, x, si, sj, sk, di, dj, dk);
    for (int64_t i = 0; i < di; ++i)
        for (int64_t j = 0; j < dj; ++j)
            for (int64_t k = 0; k < dk; ++k) {  /* residual of the system it solved */
                const int64_t m = i * si + j * sj + k * sk;
                double r = b[m] * x[m] - d0[m];
                if (k > 0) r += a[m] * x[m - sk];
                if (k < dk - 1) r += c0[m] * x[m + sk];
                CHECK(fabs(r) < 1e-12 * (fabs(d0[m]) + 10.0), "tridiag residual %g at (%lld, %lld, %lld)", r, (long long)i, (long long)j, (long long)k);
            }
    free(a); free(b); free(c); free(d); free(x); free(c0); free(d0);
}

int main(void) {
    const int64_t doms[][3] = {{1, 1, 1}, {3, 5, 2}, {17, 9, 4}, {8, 1, 3}, {1, 7, 2}, {0, 4, 2}, {4, 0, 2}};
    for (unsigned t = 0; t < sizeof doms / sizeof doms[0]; ++t)
        for (int layout = 0; layout < 2; ++layout) {
            test_lap(doms[t][0], doms[t][1], doms[t][2], layout);
            test_hdiff(doms[t][0], doms[t][1], doms[t][2], layout, 1);
            test_hdiff(doms[t][0], doms[t][1], doms[t][2], layout, 0);
            if (doms[t][2] >= 2) test_tridiag(doms[t][0], doms[t][1], doms[t][2], layout);
        }
    if (failures) {
        fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    printf("oracle/cpu_ifirst.c: all entry points clean under the sanitizers\n");
    return 0;
}
