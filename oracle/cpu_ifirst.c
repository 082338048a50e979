/*
 * ORACLE (test infrastructure, NOT product code): plain-C/OpenMP restatement of the three hot-path
 * stencils with the structure of gt4py's `gt:cpu_ifirst` backend -- I-contiguous fields
 * (layout (2,1,0), /root/reference/src/gt4py/storage/cartesian/layout_registry.py:87-94), threads
 * over (K, J) rows, I innermost; K innermost and serial for the vertical solve.
 *
 * Used (a) as a second, independent check of the numpy restatement (tests/test_oracle.py) and
 * (b) as the CPU baseline timed next to the GPU numbers (bench.py "cpu_baseline", kind "port").
 * It is NOT GridTools: the reference's gt:cpu_ifirst needs gridtools-cpp 2.3.9 headers that are not
 * in this image.  Same arithmetic as oracle/ref_numpy.py: expression trees
 * of the stencil definitions, one rounding per operation; build with -ffp-contract=off.
 *
 * Definitions restated: examples/lap_cartesian_vs_next.ipynb cell 7;
 * tests/cartesian_tests/integration_tests/multi_feature_tests/stencil_definitions.py:316-328, :219-232
 * (all under /root/reference).
 *
 * Fields are described by an origin-shifted pointer and element strides (si, sj, sk).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#pragma STDC FP_CONTRACT OFF

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* out = -4.0*in + in[-1,0] + in[1,0] + in[0,-1] + in[0,1]   (left-associative) */
void oracle_lap5_f64(const double* in, int64_t isi, int64_t isj, int64_t isk, double* out, int64_t osi,
                     int64_t osj, int64_t osk, int64_t di, int64_t dj, int64_t dk) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t k = 0; k < dk; ++k)
        for (int64_t j = 0; j < dj; ++j) {
            const double* p = in + k * isk + j * isj;
            double* q = out + k * osk + j * osj;
            for (int64_t i = 0; i < di; ++i) {
                const double* c = p + i * isi;
                double r = -4.0 * c[0];
                r = r + c[-isi];
                r = r + c[isi];
                r = r + c[-isj];
                r = r + c[isj];
                q[i * osi] = r;
            }
        }
}

/* Horizontal diffusion with flux limiter; T = field type, W = double (default float64 literals).
 * Row-blocked: per (k, j) the lap/flx/fly rows it needs are recomputed into small stack/heap
 * buffers, which is value-identical to full temporaries. */
#define HDIFF_IMPL(NAME, T)                                                                          \
    void NAME(const T* in, int64_t isi, int64_t isj, int64_t isk, T* out, int64_t osi, int64_t osj,   \
              int64_t osk, const T* cf, int64_t csi, int64_t csj, int64_t csk, int64_t di, int64_t dj, \
              int64_t dk, int limiter) {                                                             \
        _Pragma("omp parallel") {                                                                    \
            double* lapm = (double*)malloc(sizeof(double) * (size_t)(di + 2) * 3);                   \
            double* lap0 = lapm + (di + 2);                                                          \
            double* lapp = lap0 + (di + 2);                                                          \
            _Pragma("omp for collapse(2) schedule(static)")                                          \
            for (int64_t k = 0; k < dk; ++k)                                                         \
                for (int64_t j = 0; j < dj; ++j) {                                                   \
                    const T* base = in + k * isk + j * isj;                                          \
                    /* lap on rows j-1, j, j+1 for i in [-1, di] */                                  \
                    for (int r = -1; r <= 1; ++r) {                                                  \
                        double* dst = r < 0 ? lapm : (r == 0 ? lap0 : lapp);                         \
                        const T* row = base + r * isj;                                               \
                        for (int64_t i = -1; i <= di; ++i) {                                         \
                            const T* c = row + i * isi;                                              \
                            const T sum = ((c[isi] + c[-isi]) + c[isj]) + c[-isj];                   \
                            dst[i + 1] = (4.0 * (double)c[0]) - (double)sum;                         \
                        }                                                                            \
                    }                                                                                \
                    for (int64_t i = 0; i < di; ++i) {                                               \
                        const T* c = base + i * isi;                                                 \
                        double flx, flxm, fly, flym, res;                                            \
                        res = lap0[i + 2] - lap0[i + 1];                                             \
                        flx = (limiter && (res * (double)(T)(c[isi] - c[0])) > 0.0) ? 0.0 : res;     \
                        res = lap0[i + 1] - lap0[i];                                                 \
                        flxm = (limiter && (res * (double)(T)(c[0] - c[-isi])) > 0.0) ? 0.0 : res;   \
                        res = lapp[i + 1] - lap0[i + 1];                                             \
                        fly = (limiter && (res * (double)(T)(c[isj] - c[0])) > 0.0) ? 0.0 : res;     \
                        res = lap0[i + 1] - lapm[i + 1];                                             \
                        flym = (limiter && (res * (double)(T)(c[0] - c[-isj])) > 0.0) ? 0.0 : res;   \
                        const double s = ((flx - flxm) + fly) - flym;                                \
                        const double coeff = (double)cf[k * csk + j * csj + i * csi];                \
                        out[k * osk + j * osj + i * osi] = (T)((double)c[0] - (coeff * s));          \
                    }                                                                                \
                }                                                                                    \
            free(lapm);                                                                              \
        }                                                                                            \
    }

HDIFF_IMPL(oracle_hdiff_f64, double)
HDIFF_IMPL(oracle_hdiff_f32, float)

/* Thomas solve, column by column (K innermost, serial); threads over (J, I-blocks). */
void oracle_tridiag_f64(const double* inf, const double* diag, double* sup, double* rhs, double* out,
                        int64_t si, int64_t sj, int64_t sk, int64_t di, int64_t dj, int64_t dk) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < dj; ++j) {
        /* level by level over a whole row of columns keeps the I-contiguous accesses streaming */
        const int64_t r = j * sj;
        for (int64_t i = 0; i < di; ++i) {
            const int64_t o = r + i * si;
            sup[o] = sup[o] / diag[o];
            rhs[o] = rhs[o] / diag[o];
        }
        for (int64_t k = 1; k < dk; ++k) {
            const int64_t b = r + k * sk;
            for (int64_t i = 0; i < di; ++i) {
                const int64_t o = b + i * si, m = o - sk;
                const double den1 = diag[o] - (sup[m] * inf[o]);
                const double ns = sup[o] / den1;
                const double num = rhs[o] - (inf[o] * rhs[m]);
                const double den2 = diag[o] - (sup[m] * inf[o]);
                sup[o] = ns;
                rhs[o] = num / den2;
            }
        }
        {
            const int64_t b = r + (dk - 1) * sk;
            for (int64_t i = 0; i < di; ++i) out[b + i * si] = rhs[b + i * si];
        }
        for (int64_t k = dk - 2; k >= 0; --k) {
            const int64_t b = r + k * sk;
            for (int64_t i = 0; i < di; ++i) {
                const int64_t o = b + i * si;
                out[o] = rhs[o] - (sup[o] * out[o + sk]);
            }
        }
    }
}
