// fourier_resample.hpp -- host-side Fourier-method resampling of a real 1-D sequence to `num` samples.
//
// Replaces /root/reference/hss/transforms/resample.py:13-21 (Resample.__call__), whose whole body is
//     scipy.signal.resample(x.cpu(), self.num)            (scipy >= 1.11, pixi.lock pins 1.17.0)
// used for the label path of the dataset (hss/datasets/heart_sounds.py:202-207) and optionally for signals.
// Published algorithm of scipy.signal.resample for real input, window=None, domain='time':
//   X = rfft(x);  N = min(num, Nx);  Y[0 .. N/2] = X[0 .. N/2], zero above;
//   if N is even: the copied Nyquist bin is doubled when downsampling (num < Nx), halved when upsampling (Nx < num);
//   y = irfft(Y, num) * (num / Nx).
// Arbitrary lengths (a recording has 35 500 samples = 2^2 5^3 71): both transforms run as Bluestein chirp-z
// convolutions on a power-of-two radix-2 FFT, in fp64.  This is small host work per recording (labels and signals are
// a few hundred kB), like the CSV parser next to it; the GPU path of this library starts at the feature transform.
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <vector>

namespace hssfsst {
namespace resample_detail {

using cd = std::complex<double>;

inline void fft_pow2(std::vector<cd>& a, bool inverse)
{
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {              // bit reversal
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    std::vector<cd> tw(n / 2 > 0 ? n / 2 : 1);
    const double sgn = inverse ? 1.0 : -1.0;
    for (size_t k = 0; k < n / 2; ++k) {
        const double ang = sgn * 2.0 * M_PI * static_cast<double>(k) / static_cast<double>(n);
        tw[k] = cd(std::cos(ang), std::sin(ang));
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const size_t half = len >> 1, step = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; ++k) {
                const cd u = a[i + k], v = a[i + k + half] * tw[k * step];
                a[i + k] = u + v;
                a[i + k + half] = u - v;
            }
    }
    if (inverse) for (auto& v : a) v /= static_cast<double>(n);
}

// X[k] = sum_n x[n] exp(-+ 2 pi i n k / N) for any N >= 1 (forward: minus sign; inverse: plus sign, NOT scaled)
inline void dft_any(const std::vector<cd>& x, std::vector<cd>& X, bool inverse)
{
    const int64_t N = static_cast<int64_t>(x.size());
    X.assign(static_cast<size_t>(N), cd(0.0, 0.0));
    if (N == 1) { X[0] = x[0]; return; }
    size_t M = 1;
    while (M < static_cast<size_t>(2 * N - 1)) M <<= 1;
    // chirp w[n] = exp(i pi n^2 / N) (sign flipped for the inverse); n^2 reduced mod 2N keeps the angle exact
    std::vector<cd> w(static_cast<size_t>(N));
    const double sgn = inverse ? -1.0 : 1.0;
    for (int64_t n = 0; n < N; ++n) {
        const int64_t r = (n * n) % (2 * N);
        const double ang = sgn * M_PI * static_cast<double>(r) / static_cast<double>(N);
        w[static_cast<size_t>(n)] = cd(std::cos(ang), std::sin(ang));
    }
    std::vector<cd> a(M, cd(0.0, 0.0)), b(M, cd(0.0, 0.0));
    for (int64_t n = 0; n < N; ++n) a[static_cast<size_t>(n)] = x[static_cast<size_t>(n)] * std::conj(w[static_cast<size_t>(n)]);
    b[0] = w[0];
    for (int64_t n = 1; n < N; ++n) b[static_cast<size_t>(n)] = b[M - static_cast<size_t>(n)] = w[static_cast<size_t>(n)];
    fft_pow2(a, false);
    fft_pow2(b, false);
    for (size_t i = 0; i < M; ++i) a[i] *= b[i];
    fft_pow2(a, true);
    for (int64_t k = 0; k < N; ++k) X[static_cast<size_t>(k)] = a[static_cast<size_t>(k)] * std::conj(w[static_cast<size_t>(k)]);
}

}  // namespace resample_detail

// y[0 .. num) = scipy.signal.resample(x[0 .. n), num) for real x.  Returns false on bad sizes.
inline bool fourier_resample(const double* x, int64_t n, int64_t num, double* y)
{
    using resample_detail::cd;
    if (!x || !y || n < 1 || num < 1) return false;
    std::vector<cd> xin(static_cast<size_t>(n)), X;
    for (int64_t i = 0; i < n; ++i) xin[static_cast<size_t>(i)] = cd(x[i], 0.0);
    resample_detail::dft_any(xin, X, false);             // rfft = first n/2 + 1 bins of this
    const int64_t N = num < n ? num : n, nyq = N / 2 + 1;
    std::vector<cd> Y(static_cast<size_t>(num / 2 + 1), cd(0.0, 0.0));
    for (int64_t k = 0; k < nyq; ++k) Y[static_cast<size_t>(k)] = X[static_cast<size_t>(k)];
    if (N % 2 == 0) {
        if (num < n) Y[static_cast<size_t>(N / 2)] *= 2.0;
        else if (n < num) Y[static_cast<size_t>(N / 2)] *= 0.5;
    }
    // irfft(Y, num): Hermitian extension (imaginary parts of DC and of the Nyquist bin of an even num are ignored)
    std::vector<cd> Z(static_cast<size_t>(num), cd(0.0, 0.0)), z;
    Z[0] = cd(Y[0].real(), 0.0);
    for (int64_t k = 1; k <= num / 2; ++k) {
        cd v = Y[static_cast<size_t>(k)];
        if (num % 2 == 0 && k == num / 2) { Z[static_cast<size_t>(k)] = cd(v.real(), 0.0); continue; }
        Z[static_cast<size_t>(k)] = v;
        Z[static_cast<size_t>(num - k)] = std::conj(v);
    }
    resample_detail::dft_any(Z, z, true);
    const double scale = 1.0 / static_cast<double>(n);   // (1 / num) of the inverse transform * (num / n)
    for (int64_t i = 0; i < num; ++i) y[i] = z[static_cast<size_t>(i)].real() * scale;
    return true;
}

}  // namespace hssfsst
