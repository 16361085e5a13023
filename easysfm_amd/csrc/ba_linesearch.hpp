// Step-length selection of the Armijo line search that Ceres' trust-region loop runs along the LM step when the
// problem has box bounds -- the reference bounds the reference camera and the free intrinsics,
// cpp_code/src/ba.cpp:155-162, :190-194 -- [upstream trust_region_minimizer.cc DoLineSearch, line_search.cc
// ArmijoLineSearch::DoSearch + InterpolatingPolynomialMinimizingStepSize with the default CUBIC interpolation,
// polynomial.cc MinimizeInterpolatingPolynomial].  Host-side scalar code: the cost and slope at each trial come from
// ba_cost_kernel<true>.
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

namespace esfm {
namespace linesearch {

constexpr double kSufficientDecrease = 1e-4;   // Solver::Options::line_search_sufficient_function_decrease
constexpr double kMaxContraction = 1e-3;       // max_line_search_step_contraction
constexpr double kMinContraction = 0.6;        // min_line_search_step_contraction
constexpr double kMinStepSize = 1e-9;          // min_line_search_step_size
constexpr int kMaxIterations = 20;             // max_num_line_search_step_size_iterations

struct Sample {
    double x = 0.0, f = 0.0, g = 0.0;
    bool valid = false;   // value and slope both usable
};

using Poly = std::vector<double>;   // coefficient of x^i at [i]

inline double eval(const Poly &p, double x)
{
    double v = 0.0;
    for (size_t i = p.size(); i-- > 0;) v = v * x + p[i];
    return v;
}

inline Poly derivative(const Poly &p)
{
    Poly d;
    for (size_t i = 1; i < p.size(); ++i) d.push_back((double)i * p[i]);
    while (d.size() > 1 && d.back() == 0.0) d.pop_back();
    return d;
}

// Polynomial of degree 2 m - 1 through the values and slopes of m samples (complete-pivoting elimination, like the
// fullPivLu Ceres solves the same system with).  Empty on a singular system.
inline Poly interpolate(const std::vector<Sample> &s)
{
    const int n = 2 * (int)s.size();
    std::vector<std::vector<double>> A((size_t)n, std::vector<double>((size_t)n + 1, 0.0));
    for (size_t i = 0; i < s.size(); ++i) {
        double pw = 1.0;
        for (int j = 0; j < n; ++j) { A[2 * i][(size_t)j] = pw; pw *= s[i].x; }
        A[2 * i][(size_t)n] = s[i].f;
        pw = 1.0;
        for (int j = 1; j < n; ++j) { A[2 * i + 1][(size_t)j] = (double)j * pw; pw *= s[i].x; }
        A[2 * i + 1][(size_t)n] = s[i].g;
    }
    std::vector<int> col((size_t)n);
    for (int j = 0; j < n; ++j) col[(size_t)j] = j;
    for (int k = 0; k < n; ++k) {
        int br = k, bc = k;
        for (int i = k; i < n; ++i)
            for (int j = k; j < n; ++j)
                if (std::fabs(A[(size_t)i][(size_t)j]) > std::fabs(A[(size_t)br][(size_t)bc])) { br = i; bc = j; }
        if (A[(size_t)br][(size_t)bc] == 0.0) return {};
        std::swap(A[(size_t)k], A[(size_t)br]);
        if (bc != k) {
            for (int i = 0; i < n; ++i) std::swap(A[(size_t)i][(size_t)k], A[(size_t)i][(size_t)bc]);
            std::swap(col[(size_t)k], col[(size_t)bc]);
        }
        for (int i = k + 1; i < n; ++i) {
            const double f = A[(size_t)i][(size_t)k] / A[(size_t)k][(size_t)k];
            for (int j = k; j <= n; ++j) A[(size_t)i][(size_t)j] -= f * A[(size_t)k][(size_t)j];
        }
    }
    Poly c((size_t)n, 0.0);
    for (int i = n - 1; i >= 0; --i) {
        double v = A[(size_t)i][(size_t)n];
        for (int j = i + 1; j < n; ++j) v -= A[(size_t)i][(size_t)j] * c[(size_t)col[(size_t)j]];
        c[(size_t)col[(size_t)i]] = v / A[(size_t)i][(size_t)i];
    }
    return c;
}

// Real zeros of p inside [lo, hi]: p is monotone between consecutive zeros of p', so every sign change over such
// a stretch brackets exactly one zero, found by bisection.
inline std::vector<double> zeros_in(Poly p, double lo, double hi)
{
    while (p.size() > 1 && p.back() == 0.0) p.pop_back();
    std::vector<double> out;
    if (p.size() < 2) return out;
    if (p.size() == 2) {
        const double z = -p[0] / p[1];
        if (z >= lo && z <= hi) out.push_back(z);
        return out;
    }
    std::vector<double> knots = zeros_in(derivative(p), lo, hi);
    std::sort(knots.begin(), knots.end());
    knots.insert(knots.begin(), lo);
    knots.push_back(hi);
    for (size_t i = 0; i + 1 < knots.size(); ++i) {
        double a = knots[i], b = knots[i + 1];
        if (!(b > a)) continue;
        double fa = eval(p, a);
        const double fb = eval(p, b);
        if (fa == 0.0) { if (out.empty() || out.back() != a) out.push_back(a); continue; }
        if (fb == 0.0) { if (i + 2 == knots.size()) out.push_back(b); continue; }
        if ((fa < 0.0) == (fb < 0.0)) continue;
        for (int it = 0; it < 200; ++it) {
            const double m = 0.5 * (a + b);
            if (m == a || m == b) break;
            const double fm = eval(p, m);
            if (fm == 0.0) { a = b = m; break; }
            if ((fm < 0.0) == (fa < 0.0)) { a = m; fa = fm; } else b = m;
        }
        out.push_back(0.5 * (a + b));
    }
    return out;
}

// Minimiser of p over [lo, hi] among the midpoint, the ends and the critical points (polynomial.cc MinimizePolynomial;
// for a quadratic p' the candidates are those of FindQuadraticPolynomialRoots, whose "real part of a complex pair"
// Ceres also tries).
inline double argmin(const Poly &p, double lo, double hi)
{
    double bx = 0.5 * (lo + hi), bv = eval(p, bx);
    auto consider = [&](double x) { const double v = eval(p, x); if (v < bv) { bv = v; bx = x; } };
    consider(lo);
    consider(hi);
    if (p.size() <= 2) return bx;
    const Poly d = derivative(p);
    std::vector<double> z;
    if (d.size() == 2) z.push_back(-d[0] / d[1]);
    else if (d.size() == 3) {
        const double a = d[2], b = d[1], c = d[0];
        const double disc = b * b - 4.0 * a * c, sq = std::sqrt(std::fabs(disc));
        if (disc >= 0.0) {
            if (b >= 0.0) { z.push_back((-b - sq) / (2.0 * a)); z.push_back((2.0 * c) / (-b - sq)); }
            else { z.push_back((2.0 * c) / (-b + sq)); z.push_back((-b + sq) / (2.0 * a)); }
        } else z.push_back(-b / (2.0 * a));
    } else if (d.size() > 3) z = zeros_in(d, lo, hi);
    for (double x : z) if (x >= lo && x <= hi) consider(x);
    return bx;
}

// Next trial step after `current` failed the sufficient-decrease test (or could not be evaluated).
inline double next_step(const Sample &initial, const Sample &previous, const Sample &current)
{
    const double lo = kMaxContraction * current.x, hi = kMinContraction * current.x;
    const double halved = std::min(std::max(0.5 * current.x, lo), hi);
    if (!current.valid) return halved;
    std::vector<Sample> s{initial, current};
    if (previous.valid) s.push_back(previous);
    const Poly p = interpolate(s);
    return p.empty() ? halved : argmin(p, lo, hi);
}

}  // namespace linesearch
}  // namespace esfm
