/* Test infrastructure: drives oracle/fsst_oracle.c under -fsanitize=address,undefined (tests/test_oracle.py::test_oracle_under_sanitizers):
 * the shapes of the restatement tests (the seven (N, n) cases of test_c_vs_numpy_restatement), the whole feature call in its three
 * modes with and without a band, ragged / degenerate sizes, several threads.  Exactly sized heap buffers: an access one element
 * outside any of them, a signed overflow, a misaligned or NULL access aborts the run (-fno-sanitize-recover).  Prints a checksum
 * so that the run cannot be optimised away; the numbers themselves are checked by the Python tests. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int hss_oracle_dtwin(const double* w, int n, double fs, double* dw);
int hss_oracle_fsst(const double* x, int nx, double fs, const double* w, int N, double* s_re, double* s_im, double* f, double* t, double* halfdist);
int hss_oracle_band(int N, double fs, double f_lo, double f_hi, int* klo);
int hss_oracle_features(const float* x, int64_t batch, int nx, double fs, const double* w, int N, int has_band, double f_lo, double f_hi,
                        int mode, float* out, double* halfdist, int nthreads);
double hss_oracle_update_mean(double m, double x, int64_t k);
double hss_oracle_update_variance(double x, double m, double var, int64_t k);

static uint64_t rng = 0x9e3779b97f4a7c15ull;
static double rnd(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0 - 0.5; }

int main(void)
{
    static const int cases[][2] = {{128, 2000}, {128, 300}, {128, 1}, {64, 257}, {256, 400}, {100, 333}, {33, 90}, {1, 5}, {2, 3}, {7, 7}};
    double sum = 0.0;
    for (unsigned c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c) {
        const int N = cases[c][0], n = cases[c][1], nf = N / 2 + 1;
        double* w = malloc(sizeof(double) * (size_t)N);
        double* dw = malloc(sizeof(double) * (size_t)N);
        double* x = malloc(sizeof(double) * (size_t)n);
        double* sr = malloc(sizeof(double) * (size_t)nf * (size_t)n);
        double* si = malloc(sizeof(double) * (size_t)nf * (size_t)n);
        double* f = malloc(sizeof(double) * (size_t)nf);
        double* t = malloc(sizeof(double) * (size_t)n);
        double* hd = malloc(sizeof(double) * (size_t)n);
        for (int i = 0; i < N; ++i) w[i] = 0.54 - 0.46 * cos(6.283185307179586 * (i + 0.5) / N) + 0.01 * rnd();
        for (int i = 0; i < n; ++i) x[i] = rnd();
        if (hss_oracle_dtwin(w, N, 1000.0, dw) != 0) return 2;
        { const int rc3 = hss_oracle_fsst(x, n, 1000.0, w, N, sr, si, f, t, hd); if (rc3 != nf) { fprintf(stderr, "fsst = %d (N %d n %d)\n", rc3, N, n); return 3; } }
        { const int rc4 = hss_oracle_fsst(x, n, 1000.0, w, N, sr, si, NULL, NULL, NULL); if (rc4 != nf) { fprintf(stderr, "fsst(optional outputs absent) = %d (N %d n %d)\n", rc4, N, n); return 4; } }
        for (int i = 0; i < nf * n; ++i) sum += sr[i] + si[i];
        /* the whole feature call: 3 windows (ragged batch for the thread split), every mode, with and without a band */
        const int B = 3;
        float* xf = malloc(sizeof(float) * (size_t)B * (size_t)n);
        for (int i = 0; i < B * n; ++i) xf[i] = (float)rnd();
        for (int has = 0; has < 2; ++has) {
            int klo = 0;
            const int K = has ? hss_oracle_band(N, 1000.0, 25.0, 200.0, &klo) : nf;
            if (K < 0) return 5;
            for (int mode = 0; mode < 3; ++mode) {
                const size_t per = (size_t)n * (size_t)K * (mode == 1 ? 1u : 2u);
                float* out = malloc(sizeof(float) * (per * B ? per * B : 1));
                double* hb = malloc(sizeof(double) * (size_t)B * (size_t)n);
                for (int th = 1; th <= 4; th += 3)
                    if (K > 0 && hss_oracle_features(xf, B, n, 1000.0, w, N, has, 25.0, 200.0, mode, out, th == 1 ? hb : NULL, th) != 0) return 6;
                for (size_t i = 0; i < per * B; ++i) if (out[i] == out[i]) sum += out[i];
                free(out); free(hb);
            }
        }
        free(xf); free(w); free(dw); free(x); free(sr); free(si); free(f); free(t); free(hd);
    }
    double m = 0.0, v = 0.0;
    for (int64_t k = 1; k <= 1000; ++k) { const double xk = rnd(); v = hss_oracle_update_variance(xk, m, v, k); m = hss_oracle_update_mean(m, xk, k); }
    printf("oracle sanitize ok %.6e %.6e %.6e\n", sum, m, v);
    return 0;
}
