// tests/host_logic/san_driver.cpp -- the product's host logic (rocco_amd/csrc/search.cpp) under AddressSanitizer and
// UndefinedBehaviorSanitizer (CPU only: the pool has no GPU sanitizers).  One executable: the harness (search.cpp on the
// CPU evaluator), the oracle's C sources and this driver, all compiled with -fsanitize=address,undefined.  Random
// calibrations and fixed-penalty solves over several kinds of score arrays and over the device-side behaviours the
// harness can imitate (compaction at once / with slack, small tiles, noisy pilots, few and many penalties per round);
// every result is compared with the oracle's sequential calibration.  Exit code 0 and no sanitizer report = pass.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../oracle/oracle.h"

extern "C" {
int hostlogic_calibrate(const double *scores, const double *costs, double gamma, size_t n, long long target, double sum_costs,
                        int max_iter, int spec_depth, int force_exact, uint8_t *solution, double *penalty_out, double *value_out,
                        long long *count_out, long long *out_i);
int hostlogic_solve_fixed(const double *scores, const double *costs, double gamma, size_t n, double lambda, uint8_t *solution,
                          double *value_out, long long *count_out, long long *out_i);
}

static uint64_t state = 0x9E3779B97F4A7C15ULL;
static double uniform()
{
    state ^= state << 13;
    state ^= state >> 7;
    state ^= state << 17;
    return (double)(state >> 11) / 9007199254740992.0;
}

// np.sum of a constant vector (pairwise summation in blocks of 8 x unrolled 128, as the product restates it) is not restated
// here: the driver passes costs as VECTORS and sums them left to right for both sides, which is all the comparison needs
static double sum_of(const std::vector<double> &v)
{
    double s = 0.0;
    for (double x : v) s += x;
    return s;
}

int main(int argc, char **argv)
{
    const int cases = (argc > 1) ? std::atoi(argv[1]) : 160;
    const char *behaviours[][2] = {{"ROCCO_HOSTLOGIC_NOW", "1"}, {"ROCCO_HOSTLOGIC_SLACK", "0.5"}, {"ROCCO_HOSTLOGIC_TILE", "64"},
                                   {"ROCCO_HOSTLOGIC_PILOT", "0.6"}, {"ROCCO_HOSTLOGIC_POINTS", "7"}, {"ROCCO_HOSTLOGIC_POINTS", "64"},
                                   {"ROCCO_HOSTLOGIC_PILOT_ROUNDS", "1"}, {"ROCCO_HOSTLOGIC_COMPACT", "0"}};
    const int n_behaviours = (int)(sizeof(behaviours) / sizeof(behaviours[0]));
    int bad = 0;
    for (int c = 0; c < cases; ++c) {
        const int kind = c % 6;
        const size_t n = (c % 11 == 0) ? (size_t)(1 + c % 7) : (size_t)(50 + uniform() * ((c % 5 == 0) ? 70000 : 9000));
        std::vector<double> s(n);
        for (size_t i = 0; i < n; ++i) {
            const double u = uniform(), g = std::sqrt(-2.0 * std::log(uniform() + 1e-300)) * std::cos(6.283185307179586 * uniform());
            switch (kind) {
            case 0: s[i] = g; break;
            case 1: s[i] = std::floor(4.0 * u); break;                              // integer ties
            case 2: s[i] = std::round(-std::log(u + 1e-300) * 0.3 * 1e5) / 1e5; break;  // 5-decimal background
            case 3: s[i] = 1.0e6 + g; break;                                        // far from zero
            case 4: s[i] = (u < 0.97) ? 0.0 : 5.0 * uniform(); break;               // mostly exact zeros
            default: s[i] = 1e-9 * g; break;
            }
        }
        if (kind == 2) {
            for (size_t p = 20; p + 40 < n; p += 700) {
                for (size_t i = p; i < p + 5 + (size_t)(uniform() * 30); ++i) s[i] += 2.0 + 6.0 * uniform();
            }
        }
        const bool vector_costs = (c % 4 == 3) && n > 1;
        const double gamma = (c % 9 == 0) ? 0.0 : (0.25 + 2.0 * uniform());
        std::vector<double> costs(n > 1 ? n - 1 : 0, gamma);
        if (vector_costs) {
            for (double &x : costs) x = 0.2 + 1.1 * uniform();
        }
        const double total = sum_of(costs);
        const long long target = (long long)std::floor((double)n * (0.005 + 0.1 * uniform()));
        double smin = s[0], smax = s[0];
        for (double x : s) {
            smin = std::fmin(smin, x);
            smax = std::fmax(smax, x);
        }
        // what the reference's sequential calibration returns (oracle: rocco/dp.py:89-164 over rocco/_chain_dp.c:109-186)
        std::vector<uint8_t> want(n), got(n);
        double want_pen = 0.0, want_val = 0.0, got_pen = 0.0, got_val = 0.0;
        long long want_cnt = 0, got_cnt = 0, info[16] = {0};
        int evals = 0;
        if (oracle_calibrate_selection_penalty_f64(s.data(), costs.empty() ? nullptr : costs.data(), gamma, n, target, 60, total, smin, smax,
                                                   &want_pen, want.data(), &want_val, &want_cnt, &evals) != 0) {
            std::fprintf(stderr, "case %d: the oracle failed\n", c);
            return 2;
        }
        const int b = c % (n_behaviours + 1);
        if (b < n_behaviours) {
            setenv(behaviours[b][0], behaviours[b][1], 1);
        }
        const int rc = hostlogic_calibrate(s.data(), costs.empty() ? nullptr : costs.data(), gamma, n, target, total, 60, 2, (c % 13 == 5) ? 1 : 0,
                                           got.data(), &got_pen, &got_val, &got_cnt, info);
        if (b < n_behaviours) {
            unsetenv(behaviours[b][0]);
        }
        if (rc != 0 || got_pen != want_pen || got_cnt != want_cnt || got != want) {
            std::fprintf(stderr, "case %d (kind %d, n %zu, behaviour %d): rc %d penalty %.17g / %.17g count %lld / %lld\n", c, kind, n, b, rc,
                         got_pen, want_pen, got_cnt, want_cnt);
            ++bad;
        }
        // a fixed-penalty solve at the calibrated penalty and at one between the extremes
        for (double lambda : {want_pen, smin + (smax - smin) * uniform()}) {
            long long cnt = 0, ocnt = 0;
            double val = 0.0, oval = 0.0;
            std::vector<uint8_t> sol(n), osol(n);
            oracle_solve_penalized_chain_f64(s.data(), costs.empty() ? nullptr : costs.data(), gamma, n, lambda, osol.data(), &oval, &ocnt);
            const int rf = hostlogic_solve_fixed(s.data(), costs.empty() ? nullptr : costs.data(), gamma, n, lambda, sol.data(), &val, &cnt, info);
            if (rf != 0 || cnt != ocnt || sol != osol) {
                std::fprintf(stderr, "case %d fixed penalty %.17g: rc %d count %lld / %lld\n", c, lambda, rf, cnt, ocnt);
                ++bad;
            }
        }
    }
    std::printf("%d cases, %d mismatches\n", cases, bad);
    return bad == 0 ? 0 : 1;
}
