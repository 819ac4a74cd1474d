// savgol.hpp -- host-side generator of the Savitzky-Golay operator used at
// control/src/mppi:202:  uvec = savgol_filter(uvec, horizon - 1, 3, axis=1)   (scipy, mode='interp').
//
// With window n = T-1 on a length-T signal there are only two windows, x[0..T-2] and x[1..T-1]; scipy's 'interp'
// mode fits one least-squares cubic to each and evaluates it at the sample positions (the centre sample of an odd
// window is the usual convolution output).  The filter is therefore the fixed linear map  u_f = u @ S  with
// S [T][T] of rank <= 8.  Built with Gram (discrete orthonormal) polynomials on the window -- numerically a
// different route from a Vandermonde solve, so the oracle's version cross-checks it.
//
// Even windows (an ODD horizon).  The scipy of the reference's ROS era refused them; scipy >= 1.x (what the reference
// runs on today) accepts: each edge fit then covers n/2 samples, and the single interior sample j = n/2 is the
// convolution with coefficients made for the half-integer position n/2 - 1/2, which scipy.ndimage aligns on
// x[1..n] -- the right window's cubic evaluated at n/2 - 1/2.  One extra column of the basis (the polynomials at that
// position) covers it; pinned by scipy-generated fixtures (savgol_S_7 / _51 / _101).
#pragma once
#include <cmath>
#include <vector>

namespace mppi {

constexpr int kSavgolOrder = 3;

// position of output j inside its window, and which window (shift 0: x[0..n-1], 1: x[1..n]); mid = the even window's
// one half-integer position (its value comes from the basis' extra column)
inline void savgol_place(int T, int j, int& shift, int& e, bool& mid) {
    const int n = T - 1;
    mid = false;
    if (n % 2) { shift = (j <= (n - 1) / 2) ? 0 : 1; e = j - shift; return; }
    shift = (j < n / 2) ? 0 : 1;
    e = j - shift;
    mid = j == n / 2;
}

// The orthonormal basis itself, basis[d * (T-1) + i] = p_d(i), d = 0..3, i = 0..T-2, followed by FOUR more values
// basis[4 (T-1) + d] = p_d((T-1)/2 - 1/2) (used by even windows only).  The operator is
//   S[i + shift_j][j] = sum_d p_d(e_j) p_d(i)
// so  (u @ S)[j] = sum_d p_d(e_j) c_d[shift_j]  with the 8 coefficients  c_d[s] = sum_i p_d(i) u[i + s]  --
// the finalize kernel filters through those (8 (T-1) + 4 T multiply-adds per wheel instead of T^2, 32 (T-1) bytes of
// operator instead of 8 T^2).  returns false if the window T-1 is <= polyorder
inline bool savgol_basis(int T, std::vector<double>& basis) {
    const int n = T - 1;
    const int order = kSavgolOrder;
    if (n <= order) return false;
    // by modified Gram-Schmidt on z^d; the same operations run on the monomial coefficients a[d][k] of p_d(z), so that
    // p_d can be evaluated between the grid points
    std::vector<std::vector<double>> p(order + 1, std::vector<double>(n));
    double a[kSavgolOrder + 1][kSavgolOrder + 1] = {};
    const double c = 0.5 * (n - 1);
    for (int d = 0; d <= order; ++d) {
        for (int i = 0; i < n; ++i) p[d][i] = std::pow((i - c) / c, d);
        a[d][d] = 1.0;
        for (int pass = 0; pass < 2; ++pass)  // re-orthogonalise once for full double accuracy
            for (int e = 0; e < d; ++e) {
                double dot = 0.0;
                for (int i = 0; i < n; ++i) dot += p[d][i] * p[e][i];
                for (int i = 0; i < n; ++i) p[d][i] -= dot * p[e][i];
                for (int k = 0; k <= order; ++k) a[d][k] -= dot * a[e][k];
            }
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm += p[d][i] * p[d][i];
        nrm = std::sqrt(nrm);
        for (int i = 0; i < n; ++i) p[d][i] /= nrm;
        for (int k = 0; k <= order; ++k) a[d][k] /= nrm;
    }
    basis.assign((size_t)(order + 1) * n + (order + 1), 0.0);
    for (int d = 0; d <= order; ++d)
        for (int i = 0; i < n; ++i) basis[(size_t)d * n + i] = p[d][i];
    const double zm = (0.5 * n - 0.5 - c) / c;
    for (int d = 0; d <= order; ++d) {
        double v = 0.0;
        for (int k = order; k >= 0; --k) v = v * zm + a[d][k];
        basis[(size_t)(order + 1) * n + d] = v;
    }
    return true;
}

inline bool savgol_operator(int T, std::vector<double>& S) {
    std::vector<double> basis;
    if (!savgol_basis(T, basis)) return false;
    const int n = T - 1, order = kSavgolOrder;
    S.assign((size_t)T * T, 0.0);
    for (int j = 0; j < T; ++j) {
        int shift, e;
        bool mid;
        savgol_place(T, j, shift, e, mid);
        for (int i = 0; i < n; ++i) {
            double h = 0.0;   // hat-matrix entry: fitted value at the output's position from window sample i
            for (int d = 0; d <= order; ++d) h += (mid ? basis[(size_t)(order + 1) * n + d] : basis[(size_t)d * n + e]) * basis[(size_t)d * n + i];
            S[(size_t)(i + shift) * T + j] = h;
        }
    }
    return true;
}

}  // namespace mppi
