/*
 * oracle/chain_oracle.c -- exact CPU restatement of the reference chain solver.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The arithmetic (IEEE double add / subtract / compare, in this association order) follows
 *   rocco/_chain_dp.c:109-112   initial states
 *   rocco/_chain_dp.c:117-129   candidate values:  prev1 - c ;  (prev1 + s) - lambda ;
 *                                                   ((prev0 - c) + s) - lambda
 *   rocco/_chain_dp.c:133-159   pick: larger value, then fewer selected, then "stay"
 *   rocco/_chain_dp.c:167-179   terminal pick (state 1 iff strictly better, or equal with fewer)
 *   rocco/_chain_dp.c:181-186   backtrack
 * and the calibration loop follows rocco/dp.py:89-164.
 * Storage differs from the reference (two decision bits per locus packed four loci to a byte).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double val;
    long long cnt;
} path_t;

/* 1 if candidate `a` beats incumbent `b` under (value desc, count asc); ties keep `b`. */
static inline int beats(path_t a, path_t b)
{
    return (a.val > b.val) || (a.val == b.val && a.cnt < b.cnt);
}

int oracle_solve_penalized_chain_f64(const double *scores, const double *switch_costs, double gamma,
                                     size_t n, double selection_penalty, uint8_t *solution,
                                     double *value_out, long long *count_out)
{
    if (scores == NULL || n == 0) {
        return -2;
    }
    const double lam = selection_penalty;
    uint8_t *decisions = NULL; /* bit 0: from state 0 go back to state 1; bit 1: from state 1 stay in 1 */
    if (solution != NULL) {
        decisions = (uint8_t *)calloc((n + 3) / 4, 1);
        if (decisions == NULL) {
            return -1;
        }
    }

    path_t off = {0.0, 0};
    path_t on = {scores[0] - lam, 1};

    for (size_t i = 1; i < n; ++i) {
        const double c = (switch_costs != NULL) ? switch_costs[i - 1] : gamma;
        const double s = scores[i];

        const path_t leave = {on.val - c, on.cnt};                      /* 1 -> 0 */
        const path_t keep_on = {on.val + s - lam, on.cnt + 1};          /* 1 -> 1 */
        const path_t enter = {off.val - c + s - lam, off.cnt + 1};      /* 0 -> 1 */

        const int take_leave = beats(leave, off);
        const int take_enter = beats(enter, keep_on);

        if (decisions != NULL) {
            const unsigned bits = (unsigned)take_leave | ((unsigned)(!take_enter) << 1);
            decisions[i >> 2] |= (uint8_t)(bits << ((i & 3U) * 2U));
        }
        const path_t next_off = take_leave ? leave : off;
        const path_t next_on = take_enter ? enter : keep_on;
        off = next_off;
        on = next_on;
    }

    int state = beats(on, off) ? 1 : 0;
    const path_t best = state ? on : off;

    if (solution != NULL) {
        solution[n - 1] = (uint8_t)state;
        for (size_t i = n - 1; i > 0; --i) {
            const unsigned bits = (decisions[i >> 2] >> ((i & 3U) * 2U)) & 3U;
            state = (state == 0) ? (int)(bits & 1U) : (int)((bits >> 1) & 1U);
            solution[i - 1] = (uint8_t)state;
        }
        free(decisions);
    }
    if (value_out != NULL) {
        *value_out = best.val;
    }
    if (count_out != NULL) {
        *count_out = best.cnt;
    }
    return 0;
}

int oracle_calibrate_selection_penalty_f64(const double *scores, const double *switch_costs,
                                           double gamma, size_t n, long long target_count,
                                           int max_iter, double sum_costs, double score_min,
                                           double score_max, double *penalty_out, uint8_t *solution,
                                           double *value_out, long long *count_out,
                                           int *evaluations_out)
{
    if (scores == NULL || n == 0) {
        return -2;
    }
    int evals = 0;
    long long target = target_count;
    if (target < 0) {
        target = 0;
    }
    if (target > (long long)n) {
        target = (long long)n;
    }
    double value = 0.0;
    long long count = 0;
    int rc;

    if (target == (long long)n) { /* dp.py:102-108 */
        rc = oracle_solve_penalized_chain_f64(scores, switch_costs, gamma, n, 0.0, solution, &value,
                                              &count);
        if (rc != 0) {
            return rc;
        }
        *penalty_out = 0.0;
        *value_out = value;
        *count_out = count;
        if (evaluations_out) {
            *evaluations_out = 1;
        }
        return 0;
    }

    uint8_t *scratch = (uint8_t *)malloc(n);
    if (scratch == NULL) {
        return -1;
    }

    double lower = score_min - sum_costs - 1.0; /* dp.py:110 */
    double upper = score_max + sum_costs + 1.0; /* dp.py:111 */

    rc = oracle_solve_penalized_chain_f64(scores, switch_costs, gamma, n, lower, NULL, &value, &count);
    ++evals;
    while (rc == 0 && count <= target) { /* dp.py:118-125 */
        lower -= fmax(1.0, fabs(lower));
        rc = oracle_solve_penalized_chain_f64(scores, switch_costs, gamma, n, lower, NULL, &value,
                                              &count);
        ++evals;
    }
    if (rc != 0) {
        free(scratch);
        return rc;
    }

    double best_value = 0.0;
    long long best_count = 0;
    rc = oracle_solve_penalized_chain_f64(scores, switch_costs, gamma, n, upper, solution, &best_value,
                                          &best_count);
    ++evals;
    while (rc == 0 && best_count > target) { /* dp.py:132-138 */
        upper += fmax(1.0, fabs(upper));
        rc = oracle_solve_penalized_chain_f64(scores, switch_costs, gamma, n, upper, solution,
                                              &best_value, &best_count);
        ++evals;
    }
    if (rc != 0) {
        free(scratch);
        return rc;
    }

    for (int it = 0; it < max_iter; ++it) { /* dp.py:141-162 */
        const double midpoint = (lower + upper) / 2.0;
        rc = oracle_solve_penalized_chain_f64(scores, switch_costs, gamma, n, midpoint, scratch,
                                              &value, &count);
        ++evals;
        if (rc != 0) {
            free(scratch);
            return rc;
        }
        if (count > target) {
            lower = midpoint;
        } else {
            upper = midpoint;
            best_value = value;
            best_count = count;
            if (solution != NULL) {
                memcpy(solution, scratch, n);
            }
        }
    }
    free(scratch);
    *penalty_out = upper;
    *value_out = best_value;
    *count_out = best_count;
    if (evaluations_out) {
        *evaluations_out = evals;
    }
    return 0;
}

double oracle_objective_value_f64(const uint8_t *solution, const double *scores,
                                  const double *switch_costs, double gamma, size_t n)
{
    double gain = 0.0;
    double penalty = 0.0;
    for (size_t i = 0; i < n; ++i) {
        gain += scores[i] * (double)solution[i];
    }
    for (size_t i = 0; i + 1 < n; ++i) {
        const double c = (switch_costs != NULL) ? switch_costs[i] : gamma;
        const double d = (double)solution[i + 1] - (double)solution[i];
        penalty += c * fabs(d);
    }
    return -gain + penalty;
}
