/*
 * oracle/delta_oracle.c -- sequential definition of the delta-form ("fast path") evaluation.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference's forward pass (rocco/_chain_dp.c:115-165) keeps two path values prev0/prev1.
 * Only their difference matters for its decisions:
 *      delta_0 = a_0,   delta_i = clamp(delta_{i-1}, -c_{i-1}, +c_{i-1}) + a_i,   a_i = s_i - lambda
 * with  bt0[i] = [delta_{i-1} > c], bt1[i] = [delta_{i-1} >= -c]  (rocco/_chain_dp.c:133-159) and
 * terminal state [delta_{n-1} > 0] (rocco/_chain_dp.c:167-179), so the backtrack
 * (rocco/_chain_dp.c:181-186) is a backward fill over per-locus classes ONE / ZERO / COPY.
 *
 * The reference evaluates this in IEEE double on running values of magnitude P.  While those values
 * stay inside one binade [2^e, 2^(e+1)) every operation rounds to the grid u = 2^(e-52), and the
 * reference's own recursion is EXACTLY
 *      delta~_i = clamp(delta~_{i-1}, -rn_u(c), +rn_u(c)) + rn_u(s_i) + rn_u(-lambda)
 * (rocco/_chain_dp.c:120,125,127-128: each "+ s", "- lambda", "- c" is one rounding onto the grid of
 * a value that already lies on it), except where a rounding is an exact half-way tie.
 *
 * SPEC (shared with rocco_amd/csrc/chain_fast.hip):
 *  - arithmetic grid q = 2^qexp: every input of the recursion is a multiple of q, so every
 *    add/min/max is exact in double precision, the recursion is exactly associative and any
 *    parallel decomposition gives identical bits.
 *  - chunks of ORACLE_CHUNK = 32 loci carry a binade code (oracle_binade_map): CLEAN chunks use the
 *    reference's grid u = 2^(e-52) >= q:  a = rn_u(s) + rn_u(-lambda), c = rn_u(c); a step whose
 *    rn_u(s) (or rn_u(c_j) for a cost vector) is a half-way tie adds weight u.  HAZARD chunks
 *    (running value near a power of two, binade unknown, u < q, or rn_u(-lambda) / rn_u(gamma) is a
 *    tie) use a = rn_q(s - lambda), c = rn_q(c) and every step adds weight 4 hb + q, hb = 2^(e+2-53).
 *  - tolerance tau_j = (sum of step weights since the last provable clear clamp) + (9 hb + 2 q if
 *    the chunk is a hazard chunk).  A locus is certain iff tau_j == 0 or ||delta_j| - c_j| > tau_j,
 *    and tau_j <= ORACLE_GUARD.  With tau_j = 0 even exact ties are decided: the reference breaks
 *    value ties by selected count (rocco/_chain_dp.c:133-134,147-148,167-168) and its state-1 path
 *    always holds at least one more selected locus than its state-0 path, so delta == +c keeps the
 *    state (COPY), delta == -c switches (ZERO) and a terminal tie ends in state 0.
 *  - provable clear clamp: inside a chunk two extreme chains start from the clamp bounds (+c, -c);
 *    a locus is "known" once they agree; a known locus with |delta| - c > ORACLE_GUARD is a provable
 *    clear clamp and resets the weight sum; m_j = j - 1 - (last provable clear clamp before j).
 *  - without a map every chunk is a hazard chunk with e from the global bound
 *    Pb = 2 ((P16 + npos) / 16 + cmax + sabs + |lambda| + 1), P16 = sum floor(16 max(a_i, 0)).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>

enum { CLS_ZERO = 0, CLS_COPY = 1, CLS_ONE = 2 };

static inline double grid_round(double x, double magic)
{
    volatile double t = x + magic; /* volatile: keep the two roundings exactly as written */
    return t - magic;
}

static inline double clampd(double x, double c) { return fmin(fmax(x, -c), c); }

static void fill_backward(const uint8_t *cls, size_t n, uint8_t *solution, long long *count)
{
    uint8_t state = 0; /* cls[n-1] is never COPY (terminal rule) */
    long long total = 0;
    for (size_t j = n; j-- > 0;) {
        if (cls[j] != CLS_COPY) {
            state = (uint8_t)(cls[j] == CLS_ONE);
        }
        if (solution != NULL) {
            solution[j] = state;
        }
        total += state;
    }
    *count = total;
}

int oracle_global_exponent(const double *scores, size_t n, double lambda, int qexp, double cmax,
                           double sabs, long long *p16_out, long long *npos_out)
{
    const double magic = ldexp(1.5, 52 + qexp);
    long long p16 = 0, npos = 0;
    for (size_t j = 0; j < n; ++j) {
        const double a = grid_round(scores[j] - lambda, magic);
        if (a > 0.0) {
            p16 += (long long)(16.0 * a);
            ++npos;
        }
    }
    if (p16_out) *p16_out = p16;
    if (npos_out) *npos_out = npos;
    const double pb = 2.0 * ((double)(p16 + npos) * 0.0625 + cmax + sabs + fabs(lambda) + 1.0);
    return ilogb(pb);
}

/* Per-chunk arithmetic mode derived from the binade code. */
typedef struct {
    int clean;      /* 1: reference grid u, 0: hazard             */
    double magic_u; /* clean: 1.5 * 2^e                           */
    double half_u;  /* clean: u / 2                               */
    double w_step;  /* hazard: weight per step                    */
    double w_tie;   /* clean: weight of a rounding-tie step (= u) */
    double base;    /* hazard: 9 hb + 2 q, clean: 0               */
    double nlam;    /* clean: rn_u(-lambda)                       */
} chunk_mode;

/* Smallest exponent a hazard chunk may price its roundings with: the reference forms (value + s) before it
 * subtracts lambda (rocco/_chain_dp.c:120,125,127-128), so its intermediates leave the stay-off value P0 by up to
 * c + (c + |s - lambda|) + |s| <= 2 cmax + 2 sabs + |lambda|, whatever P0 is (P0 = 0 at the head of a chromosome). */
int oracle_hazard_floor(double cmax, double sabs, double lambda)
{
    return ilogb(2.0 * cmax + 2.0 * sabs + fabs(lambda) + 2.0);
}

static chunk_mode make_mode(int code, int force_hazard, int e_global, int qexp, double lambda,
                            double gamma_raw, int has_cost_vector, double cmax, double sabs)
{
    chunk_mode m;
    int e = (code == ORACLE_MAP_NONE) ? e_global : (code & 0x7F) - ORACLE_MAP_BIAS;
    int hazard = force_hazard || (code == ORACLE_MAP_NONE) || (code & 0x80);
    if (code != ORACLE_MAP_NONE && (code & 0x80)) {
        const int e_floor = oracle_hazard_floor(cmax, sabs, lambda);
        e = (e > e_floor) ? e : e_floor;
    }
    const double q = ldexp(1.0, qexp);
    if (!hazard && e - 52 < qexp) {
        hazard = 1; /* reference grid finer than the arithmetic grid */
    }
    m.magic_u = ldexp(1.5, e);
    m.half_u = ldexp(1.0, e - 53);
    m.nlam = grid_round(-lambda, m.magic_u);
    if (!hazard) {
        if (fabs(-lambda - m.nlam) == m.half_u) {
            hazard = 1; /* -lambda rounds as a tie on this grid */
        }
        if (!has_cost_vector && fabs(gamma_raw - grid_round(gamma_raw, m.magic_u)) == m.half_u) {
            hazard = 1;
        }
    }
    const double hb = ldexp(1.0, e + 2 - 53);
    m.clean = !hazard;
    m.w_step = 4.0 * hb + q;
    m.w_tie = 2.0 * m.half_u;
    m.base = hazard ? (9.0 * hb + 2.0 * q) : 0.0;
    return m;
}

static inline double cost_on_grid(const chunk_mode *m, double c_raw, double magic_q)
{
    return m->clean ? grid_round(c_raw, m->magic_u) : grid_round(c_raw, magic_q);
}

/* a_j and the weight of step j on the chunk's grid (c_raw_prev = cost of the clamp of this step) */
static inline void step_inputs(const chunk_mode *m, double s, double lambda, double c_raw_prev, double magic_q,
                               int cost_is_vector, double *a, double *w)
{
    if (m->clean) {
        const double rs = grid_round(s, m->magic_u);
        *a = rs + m->nlam;
        *w = (fabs(s - rs) == m->half_u) ? m->w_tie : 0.0;
        if (cost_is_vector && fabs(c_raw_prev - grid_round(c_raw_prev, m->magic_u)) == m->half_u) {
            *w += m->w_tie;
        }
    } else {
        *a = grid_round(s - lambda, magic_q);
        *w = m->w_step;
    }
}

int oracle_delta_chain_f64(const double *scores, const double *switch_costs, double gamma, size_t n,
                           double selection_penalty, int qexp, double cmax, double sabs,
                           const uint8_t *emap, uint8_t *solution, oracle_delta_stats *stats)
{
    if (scores == NULL || n == 0 || stats == NULL) {
        return -2;
    }
    uint8_t *cls = (uint8_t *)malloc(n);
    if (cls == NULL) {
        return -1;
    }
    const double lam = selection_penalty;
    const double magic_q = ldexp(1.5, 52 + qexp);
    const int e_global = oracle_global_exponent(scores, n, lam, qexp, cmax, sabs, NULL, NULL);
    const int vec = switch_costs != NULL;
    double delta = 0.0, up = 0.0, dn = 0.0, wacc = 0.0;
    long long last_clear = -1;
    long long uncertain = 0, effect = 0, max_run = 0;
    int overflow = 0;
    chunk_mode mode = make_mode(ORACLE_MAP_NONE, 0, e_global, qexp, lam, gamma, vec, cmax, sabs);

    for (size_t j = 0; j < n; ++j) {
        if (j % ORACLE_CHUNK == 0) {
            mode = make_mode(emap ? emap[j / ORACLE_CHUNK] : ORACLE_MAP_NONE, 0, e_global, qexp, lam, gamma, vec, cmax, sabs);
        }
        double a, w;
        const double c_raw_prev = (j == 0) ? 0.0 : (vec ? switch_costs[j - 1] : gamma);
        step_inputs(&mode, scores[j], lam, c_raw_prev, magic_q, vec && j > 0, &a, &w);
        if (j == 0) {
            delta = up = dn = a;
        } else {
            const double c_prev = cost_on_grid(&mode, c_raw_prev, magic_q);
            delta = clampd(delta, c_prev) + a;
            if (j % ORACLE_CHUNK == 0) { /* extremes restart at every chunk boundary */
                up = c_prev + a;
                dn = -c_prev + a;
            } else {
                up = clampd(up, c_prev) + a;
                dn = clampd(dn, c_prev) + a;
            }
        }
        wacc += w;
        const long long m = (long long)j - 1 - last_clear;
        const double tau = wacc + mode.base;
        if (tau > 0.0 && m > max_run) { /* diagnostic: longest run behind a locus that carries tolerance */
            max_run = m;
        }
        int certain;
        uint8_t k;
        if (j + 1 < n) {
            const double cj = cost_on_grid(&mode, vec ? switch_costs[j] : gamma, magic_q);
            const double e = fabs(delta) - cj;
            certain = (tau <= ORACLE_GUARD) && (tau == 0.0 || e > tau || e < -tau);
            k = (delta > cj) ? CLS_ONE : ((delta <= -cj) ? CLS_ZERO : CLS_COPY);
            if (up == dn && e > ORACLE_GUARD) {
                last_clear = (long long)j;
                wacc = 0.0;
            }
        } else {
            const double e = fabs(delta);
            certain = (tau <= ORACLE_GUARD) && (tau == 0.0 || e > tau);
            k = (delta > 0.0) ? CLS_ONE : CLS_ZERO;
        }
        if (tau > ORACLE_GUARD) {
            overflow = 1;
        }
        if (!certain) {
            ++uncertain;
            effect += m + 1;
        }
        cls[j] = k;
    }
    fill_backward(cls, n, solution, &stats->count);
    stats->uncertain = uncertain;
    stats->effect = effect;
    stats->max_run = max_run;
    stats->overflow = overflow;
    free(cls);
    return 0;
}

int oracle_delta_window_f64(const double *scores, const double *switch_costs, double gamma, size_t n,
                            double lambda_lo, double lambda_hi, int qexp, double cmax, double sabs,
                            const uint8_t *emap, uint8_t *solution, oracle_window_stats *stats,
                            oracle_window_diff *diffs, int diff_capacity)
{
    if (scores == NULL || n == 0 || stats == NULL || !(lambda_lo <= lambda_hi)) {
        return -2;
    }
    uint8_t *lo_cls = (uint8_t *)malloc(n);
    uint8_t *hi_cls = (uint8_t *)malloc(n);
    if (lo_cls == NULL || hi_cls == NULL) {
        free(lo_cls);
        free(hi_cls);
        return -1;
    }
    const double magic_q = ldexp(1.5, 52 + qexp);
    const int e_global = oracle_global_exponent(scores, n, lambda_lo, qexp, cmax, sabs, NULL, NULL);
    const int vec = switch_costs != NULL;
    /* chain 0 uses lambda_lo (larger delta), chain 1 uses lambda_hi (smaller delta) */
    double d[2] = {0.0, 0.0}, up[2] = {0.0, 0.0}, dn[2] = {0.0, 0.0};
    const double lam[2] = {lambda_lo, lambda_hi};
    double wacc = 0.0;
    long long last_clear = -1, max_run = 0, n_diff = 0;
    int adjacent = 1, overflow = 0;
    chunk_mode mode[2];
    mode[0] = make_mode(ORACLE_MAP_NONE, 0, e_global, qexp, lam[0], gamma, vec, cmax, sabs);
    mode[1] = mode[0];

    for (size_t j = 0; j < n; ++j) {
        if (j % ORACLE_CHUNK == 0) {
            const int code = emap ? emap[j / ORACLE_CHUNK] : ORACLE_MAP_NONE;
            mode[0] = make_mode(code, 0, e_global, qexp, lam[0], gamma, vec, cmax, sabs);
            mode[1] = make_mode(code, 0, e_global, qexp, lam[1], gamma, vec, cmax, sabs);
            if (mode[0].clean != mode[1].clean) { /* one penalty ties on this grid: both hazard */
                mode[0] = make_mode(code, 1, e_global, qexp, lam[0], gamma, vec, cmax, sabs);
                mode[1] = make_mode(code, 1, e_global, qexp, lam[1], gamma, vec, cmax, sabs);
            }
        }
        double wmax = 0.0;
        const double c_raw_prev = (j == 0) ? 0.0 : (vec ? switch_costs[j - 1] : gamma);
        for (int k = 0; k < 2; ++k) {
            double a, w;
            step_inputs(&mode[k], scores[j], lam[k], c_raw_prev, magic_q, vec && j > 0, &a, &w);
            if (j == 0) {
                d[k] = up[k] = dn[k] = a;
            } else {
                const double c_prev = cost_on_grid(&mode[k], c_raw_prev, magic_q);
                d[k] = clampd(d[k], c_prev) + a;
                if (j % ORACLE_CHUNK == 0) {
                    up[k] = c_prev + a;
                    dn[k] = -c_prev + a;
                } else {
                    up[k] = clampd(up[k], c_prev) + a;
                    dn[k] = clampd(dn[k], c_prev) + a;
                }
            }
            wmax = fmax(wmax, w);
        }
        wacc += wmax;
        const long long m = (long long)j - 1 - last_clear;
        const double tau = wacc + mode[0].base;
        if (tau > 0.0 && m > max_run) {
            max_run = m;
        }
        const double d_lo = d[0], d_hi = d[1];
        double cj = 0.0;
        uint8_t lo, hi;
        if (j + 1 < n) {
            cj = cost_on_grid(&mode[0], vec ? switch_costs[j] : gamma, magic_q);
            /* lowest class the reference could take anywhere in the zone (a tie may go either way) */
            if (d_hi + cj <= tau) {
                lo = CLS_ZERO;
            } else if (d_hi - cj > tau) {
                lo = CLS_ONE;
            } else {
                lo = CLS_COPY;
            }
            /* highest class (tau == 0: the exact tie rules, delta == c is COPY, delta == -c is ZERO) */
            if ((tau > 0.0) ? (d_lo - cj >= -tau) : (d_lo - cj > 0.0)) {
                hi = CLS_ONE;
            } else if ((tau > 0.0) ? (d_lo + cj < -tau) : (d_lo + cj <= 0.0)) {
                hi = CLS_ZERO;
            } else {
                hi = CLS_COPY;
            }
            /* clear for every lambda in the zone; both chains must be exactly known here */
            if (up[0] == dn[0] && up[1] == dn[1] && (d_hi - cj > ORACLE_GUARD || -d_lo - cj > ORACLE_GUARD)) {
                last_clear = (long long)j;
                wacc = 0.0;
            }
        } else {
            lo = (d_hi > tau) ? CLS_ONE : CLS_ZERO;
            hi = ((tau > 0.0) ? (d_lo >= -tau) : (d_lo > 0.0)) ? CLS_ONE : CLS_ZERO;
        }
        if (tau > ORACLE_GUARD) { /* tolerance model no longer valid */
            overflow = 1;
        }
        if (lo != hi) {
            /* decision boundary between the two classes: -c (ZERO|COPY), +c (COPY|ONE), 0 (terminal) */
            const double bound = (j + 1 < n) ? ((hi == CLS_ONE) ? cj : -cj) : 0.0;
            if (diffs != NULL && n_diff < diff_capacity) {
                diffs[n_diff].locus = (long long)j;
                diffs[n_diff].margin_lo = d_lo - bound;
                diffs[n_diff].margin_hi = d_hi - bound;
                diffs[n_diff].run = m;
                diffs[n_diff].cls_lo = lo;
                diffs[n_diff].cls_hi = hi;
            }
            ++n_diff;
            if (j + 1 < n && (int)hi - (int)lo != 1) {
                adjacent = 0;
            }
        }
        lo_cls[j] = lo;
        hi_cls[j] = hi;
    }
    fill_backward(hi_cls, n, NULL, &stats->count_hi);
    fill_backward(lo_cls, n, solution, &stats->count_lo);
    stats->n_diff = n_diff;
    stats->diff_adjacent = adjacent;
    stats->overflow = overflow;
    stats->max_run = max_run;
    free(lo_cls);
    free(hi_cls);
    return 0;
}

uint8_t oracle_binade_code(double p0_lo, double p0_hi, double margin)
{
    const double top = fmax(p0_hi, 1.0);
    int e = ilogb(top);
    if (e > 60) {
        e = 60;
    }
    int clean = 0;
    if (p0_lo > 0.0) {
        const double lo_edge = ldexp(1.0, e), hi_edge = ldexp(1.0, e + 1);
        clean = (p0_lo - lo_edge > margin) && (hi_edge - p0_hi > margin);
    }
    return (uint8_t)((clean ? 0 : 0x80) | (e + ORACLE_MAP_BIAS));
}

/* Binade map at penalty lambda_ref: the reference's stay-off value P0_j = sum_{i<j} max(0, delta_i - c_i)
 * (rocco/_chain_dp.c:133-145 in delta form) is tracked along the chromosome; a chunk whose P0 range
 * keeps a distance > margin from every power of two gets the clean code BIAS + e, every other
 * chunk the hazard code 0x80 | (BIAS + e).  The recursion used here is the hazard-mode one. */
int oracle_binade_map(const double *scores, const double *switch_costs, double gamma, size_t n,
                      double lambda_ref, int qexp, double margin, uint8_t *emap_out)
{
    if (scores == NULL || n == 0 || emap_out == NULL) {
        return -2;
    }
    const double magic_q = ldexp(1.5, 52 + qexp);
    double delta = 0.0, p0 = 0.0, p0_chunk_start = 0.0;
    for (size_t j = 0; j < n; ++j) {
        const double a = grid_round(scores[j] - lambda_ref, magic_q);
        if (j % ORACLE_CHUNK == 0) {
            p0_chunk_start = p0;
        }
        if (j == 0) {
            delta = a;
        } else {
            const double c_prev = grid_round(switch_costs ? switch_costs[j - 1] : gamma, magic_q);
            p0 += fmax(0.0, delta - c_prev); /* value of prev0 after step j */
            delta = clampd(delta, c_prev) + a;
        }
        if (j % ORACLE_CHUNK == ORACLE_CHUNK - 1 || j + 1 == n) {
            emap_out[j / ORACLE_CHUNK] = oracle_binade_code(p0_chunk_start, p0, margin);
        }
    }
    return 0;
}
