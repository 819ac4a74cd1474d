// savgol.hpp -- host-side generator of the Savitzky-Golay operator used at
// control/src/mppi:202:  uvec = savgol_filter(uvec, horizon - 1, 3, axis=1)   (scipy, mode='interp').
//
// With window = T-1 (odd) on a length-T signal there are only two windows, x[0..T-2] and
// x[1..T-1]; scipy's 'interp' mode fits one least-squares cubic to each and evaluates it at
// the sample positions (the centre sample of a window is the usual convolution output).
// The filter is therefore the fixed linear map  u_f = u @ S  with S [T][T] of rank <= 8.
// Built with Gram (discrete orthonormal) polynomials on the window -- numerically a
// different route from a Vandermonde solve, so the oracle's version cross-checks it.
#pragma once
#include <cmath>
#include <vector>

namespace mppi {

constexpr int kSavgolOrder = 3;

// The orthonormal basis itself, basis[d * (T-1) + i] = p_d(i), d = 0..3, i = 0..T-2: the operator is
//   S[i + shift_j][j] = sum_d p_d(e_j) p_d(i),   e_j = j (j <= half) | j - 1,  shift_j = 0 | 1,  half = (T - 2) / 2
// so  (u @ S)[j] = sum_d p_d(e_j) c_d[shift_j]  with the 8 coefficients  c_d[s] = sum_i p_d(i) u[i + s]  --
// the finalize kernel filters through those (8 (T-1) + 4 T multiply-adds per wheel instead of T^2, 32 (T-1) bytes of
// operator instead of 8 T^2).  returns false if the window T-1 is even or <= polyorder (scipy of the reference's era raises)
inline bool savgol_basis(int T, std::vector<double>& basis) {
    const int n = T - 1;
    const int order = kSavgolOrder;
    if (n <= order || (n % 2) == 0) return false;
    // by modified Gram-Schmidt on z^d
    std::vector<std::vector<double>> p(order + 1, std::vector<double>(n));
    const double c = 0.5 * (n - 1);
    for (int d = 0; d <= order; ++d) {
        for (int i = 0; i < n; ++i) p[d][i] = std::pow((i - c) / c, d);
        for (int pass = 0; pass < 2; ++pass)  // re-orthogonalise once for full double accuracy
            for (int e = 0; e < d; ++e) {
                double dot = 0.0;
                for (int i = 0; i < n; ++i) dot += p[d][i] * p[e][i];
                for (int i = 0; i < n; ++i) p[d][i] -= dot * p[e][i];
            }
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm += p[d][i] * p[d][i];
        nrm = std::sqrt(nrm);
        for (int i = 0; i < n; ++i) p[d][i] /= nrm;
    }
    basis.assign((size_t)(order + 1) * n, 0.0);
    for (int d = 0; d <= order; ++d)
        for (int i = 0; i < n; ++i) basis[(size_t)d * n + i] = p[d][i];
    return true;
}

inline bool savgol_operator(int T, std::vector<double>& S) {
    std::vector<double> basis;
    if (!savgol_basis(T, basis)) return false;
    const int n = T - 1, order = kSavgolOrder;
    auto pd = [&](int d, int i) { return basis[(size_t)d * n + i]; };
    // hat matrix H[e][j] = sum_d p_d(e) p_d(j): fitted value at window position e from sample j
    const int half = (n - 1) / 2;
    S.assign((size_t)T * T, 0.0);
    for (int j = 0; j < T; ++j) {
        const int e = (j <= half) ? j : j - 1;    // position inside its window
        const int shift = (j <= half) ? 0 : 1;    // left window starts at 0, right window at 1
        for (int i = 0; i < n; ++i) {
            double h = 0.0;
            for (int d = 0; d <= order; ++d) h += pd(d, e) * pd(d, i);
            S[(size_t)(i + shift) * T + j] = h;
        }
    }
    return true;
}

}  // namespace mppi
