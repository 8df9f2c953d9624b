// rocco_amd/csrc/whittaker_host.cpp -- the LDL^T factor of the cross-fit Whittaker system (baseline_backend.c:105-140, bands
// 198-224) walked on the HOST.
//
// The factor depends on the penalty and the length only, never on data; its recurrence is strictly sequential with two
// divisions on the dependent path of every step.  One GPU lane walks it at ~190 ns per step (1.05 s for chr1 at 50 bp,
// whittaker_factor_kernel); a host core at ~10 ns.  Both round every operation as IEEE double without contraction, so
// the entries are the same bits (tests/test_gpu_baseline.py builds both and compares the baselines).  The two parities
// are independent: two host threads.  The device keeps the table; the solves that read it stay on the device.
#include "kernels.h"

#include <thread>
#include <vector>

namespace rocco {

namespace {

inline double host_band_a0(long long i, long long n, int parity, double lambda)
{
    const double w = ((i & 1LL) == (long long)parity) ? 1.0 : 0.0;
    if (i == 0 || i == n - 1) {
        return w + lambda;
    }
    if (i == 1 || i == n - 2) {
        return w + (5.0 * lambda);
    }
    return w + (6.0 * lambda);
}

inline double host_band_a1(long long i, long long n, double lambda) { return (i == 0 || i == n - 2) ? (-2.0 * lambda) : (-4.0 * lambda); }

// entries first .. n-1 of one parity into out (d | l1 | l2, each n - first doubles); `seed`: entries first-2 and first-1
// (d, l1, l2 each) when first >= 4
void walk(long long n, double lambda, int parity, long long first, const double *seed, double *out)
{
    const long long m = n - first;
    double *d = out, *l1 = out + m, *l2 = out + 2 * m;
    double d_m2, l2_m2, d_m1, l1_m1, l2_m1;
    long long i0 = first;
    if (first >= 4) {
        d_m2 = seed[0];
        l2_m2 = seed[2];
        d_m1 = seed[3];
        l1_m1 = seed[4];
        l2_m1 = seed[5];
    } else {
        d_m2 = host_band_a0(0, n, parity, lambda);
        const double l1_m2 = host_band_a1(0, n, lambda) / d_m2;
        l2_m2 = lambda / d_m2;
        d[0] = d_m2;
        l1[0] = l1_m2;
        l2[0] = l2_m2;
        d_m1 = host_band_a0(1, n, parity, lambda) - ((l1_m2 * l1_m2) * d_m2);
        l1_m1 = (host_band_a1(1, n, lambda) - ((l2_m2 * d_m2) * l1_m2)) / d_m1;
        l2_m1 = (n > 3) ? (lambda / d_m1) : 0.0;
        d[1] = d_m1;
        l1[1] = l1_m1;
        l2[1] = l2_m1;
        i0 = 2;
    }
    for (long long i = i0; i < n; ++i) {
        double t1 = ((l1_m1 * l1_m1) * d_m1);
        const double t2 = ((l2_m2 * l2_m2) * d_m2);
        const double di = host_band_a0(i, n, parity, lambda) - t1 - t2;
        double l1i = 0.0, l2i = 0.0;
        if (i <= n - 2) {
            t1 = ((l2_m1 * d_m1) * l1_m1);
            l1i = (host_band_a1(i, n, lambda) - t1) / di;
        }
        if (i <= n - 3) {
            l2i = lambda / di;
        }
        d[i - first] = di;
        l1[i - first] = l1i;
        l2[i - first] = l2i;
        d_m2 = d_m1;
        l2_m2 = l2_m1;
        d_m1 = di;
        l1_m1 = l1i;
        l2_m1 = l2i;
    }
}

}  // namespace

int build_whittaker_factor_on_host(size_t cap, double penalty_lambda, double *factor_dev, hipStream_t stream,
                                   const double *old_factor_dev, size_t old_cap)
{
    if (cap < 4) {
        return ROCCO_HIP_EINVAL;
    }
    // a longer factor of the same penalty continues the old one: its entries 0 .. old_cap-3 carry over
    long long first = 0;
    double seed[2][6] = {{0.0}};
    if (old_factor_dev != nullptr && old_cap >= 8 && old_cap < cap) {
        first = (long long)old_cap - 2;
        for (int a = 0; a < 6; ++a) {
            ROCCO_HIP_TRY(hipMemcpyAsync(factor_dev + (size_t)a * cap, old_factor_dev + (size_t)a * old_cap, (size_t)first * sizeof(double),
                                         hipMemcpyDeviceToDevice, stream));
            // entries first-2, first-1 of array a (parity a / 3, component a % 3)
            double two[2];
            ROCCO_HIP_TRY(hipMemcpyAsync(two, old_factor_dev + (size_t)a * old_cap + (size_t)(first - 2), 2 * sizeof(double),
                                         hipMemcpyDeviceToHost, stream));
            ROCCO_HIP_TRY(hipStreamSynchronize(stream));
            seed[a / 3][a % 3] = two[0];
            seed[a / 3][3 + a % 3] = two[1];
        }
    }
    const long long m = (long long)cap - first;
    std::vector<double> host((size_t)6 * (size_t)m);
    std::thread other([&]() { walk((long long)cap, penalty_lambda, 1, first, seed[1], host.data() + (size_t)3 * (size_t)m); });
    walk((long long)cap, penalty_lambda, 0, first, seed[0], host.data());
    other.join();
    for (int a = 0; a < 6; ++a) {
        ROCCO_HIP_TRY(hipMemcpyAsync(factor_dev + (size_t)a * cap + (size_t)first, host.data() + (size_t)a * (size_t)m, (size_t)m * sizeof(double),
                                     hipMemcpyHostToDevice, stream));
    }
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));  // (`host` goes away with this call)
    return ROCCO_HIP_OK;
}

}  // namespace rocco
