// TEST INFRASTRUCTURE (host): the closed-form chain of po_beam2d_reg.hip against the reference's serial recurrence
//   x_t = logaddexp(p_{t-1} + ya_t, x_{t-1} + yb_t)       (PrefixTree.h:518-531, Log.h:17-23)
// closed form: B_t = sum_{s<=t} yb_s, c_s = p_{s-1} + ya_s - B_s, x_t = B_t + m + log(exp(seed - m) + sum_{s<=t} exp(c_s - m)).
// Prints the largest |difference| against a long-double evaluation of the recurrence for (a) the serial chain in double with
// glibc's exp / log (what the reference computes), (b) the serial chain with the device's table-driven logaddexp, (c) the
// closed form with the device's ex() / lg().  Also ex() / lg() alone against expl / logl.
//   g++ -O2 -ffp-contract=off -I tools/simt_emu scripts/check_chain_scan.cpp -o /tmp/check_chain_scan && /tmp/check_chain_scan
#include <hip/hip_runtime.h>
#include "../poreover_amd/csrc/po_device.h"
#include <random>
static long double lae_l(long double a, long double b) {
    if (a == -INFINITY && b == -INFINITY) return -INFINITY;
    const long double hi = a > b ? a : b, lo = a > b ? b : a;
    return hi + log1pl(expl(lo - hi));
}
static double lae_g(double x1, double x2) {   // Log.h as written, glibc
    const double hi = x1 >= x2 ? x1 : x2, d = x1 >= x2 ? x2 - x1 : x1 - x2;
    const double z = 1.0 + exp(d);
    return hi + ((z > 0) ? log(z) : -INFINITY);
}
int main(int argc, char** argv) {
    PoLaeTables T;
    for (int i = 0; i < 64; ++i) { T.exp_t[i][0] = PO_EXP_T[i][0]; T.exp_t[i][1] = PO_EXP_T[i][1]; }
    for (int i = 0; i < 65; ++i) for (int q = 0; q < 3; ++q) T.log_t[i][q] = PO_LOG_T[i][q];
    const PoLaeFast lae{&T};
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    {   // ex, lg alone
        double me = 0, ml = 0;
        for (long it = 0; it < 4000000; ++it) {
            const double d = -700.0 * U(rng) * U(rng);
            const long double r = expl((long double)d);
            const double e = lae.ex(d);
            const double rel = (double)fabsl(((long double)e - r) / r);
            if (rel > me) me = rel;
            const double S = exp(-600.0 * U(rng)) * (1 + 31 * U(rng));
            const double g = lae.lg(S);
            const double ab = (double)fabsl((long double)g - logl((long double)S));
            const double rl = ab / fmax(1.0, fabs(g));
            if (rl > ml) ml = rl;
        }
        printf("ex: max relative error %.3g; lg: max error relative to max(1, |log S|) %.3g\n", me, ml);
        printf("ex(-inf) %g ex(nan) %g ex(-701) %g ex(0) %.17g lg(0) %g lg(1) %g lg(2) %.17g lg(0.5) %.17g\n", lae.ex(-INFINITY), lae.ex(NAN), lae.ex(-701.0), lae.ex(0.0), lae.lg(0.0), lae.lg(1.0), lae.lg(2.0), lae.lg(0.5));
    }
    const long N = argc > 1 ? atol(argv[1]) : 200000;
    double dg = 0, df = 0, dc = 0, dcf = 0;
    long nguard = 0;
    for (long it = 0; it < N; ++it) {
        const int n = 4 + (int)(U(rng) * 28);
        double p[33], ya[32], yb[32];
        const double base = -2000.0 * U(rng);
        // a parent's alpha over the window (a ridge), log-softmax-like emissions
        const double peak = U(rng) * n, width = 1 + 6 * U(rng);
        for (int t = 0; t < n; ++t) {
            p[t] = base - (t - peak) * (t - peak) / width * (0.5 + U(rng)) - 3 * U(rng);
            if (U(rng) < 0.03) p[t] = -INFINITY;
            double l[5], s = 0;
            for (int c = 0; c < 5; ++c) { l[c] = 6 * U(rng) * U(rng) * (U(rng) < 0.5 ? 3 : 1); s += exp(l[c]); }
            ya[t] = l[0] - log(s); yb[t] = l[4] - log(s);
        }
        const double seed = (U(rng) < 0.3) ? base - 20 * U(rng) : -INFINITY;
        long double xl = seed;
        double xg = seed, xf = seed;
        // closed form
        double B[32], c[32], m = seed, run = 0;
        for (int t = 0; t < n; ++t) { run += yb[t]; B[t] = run; c[t] = (p[t] + ya[t]) - B[t]; m = fmax(m, c[t]); }
        bool bad = seed > -INFINITY && seed - m < -600.0;
        for (int t = 0; t < n; ++t) bad = bad || (c[t] > -INFINITY && c[t] - m < -600.0);
        if (bad) { ++nguard; continue; }   // (the kernel runs the serial chain for these)
        double S = lae.ex(seed - m);
        for (int t = 0; t < n; ++t) {
            xl = lae_l((long double)p[t] + ya[t], xl + yb[t]);
            xg = lae_g(p[t] + ya[t], xg + yb[t]);
            xf = lae(p[t] + ya[t], xf + yb[t]);
            S += lae.ex(c[t] - m);
            const double xc = (B[t] + m) + lae.lg(S);
            if (xl == -INFINITY) { if (xc != -INFINITY || xg != -INFINITY) printf("inf mismatch\n"); continue; }
            dg = fmax(dg, (double)fabsl(xg - xl)); df = fmax(df, (double)fabsl(xf - xl)); dc = fmax(dc, (double)fabsl(xc - xl));
            dcf = fmax(dcf, fabs(xc - xf));
        }
    }
    printf("%ld chains (%ld beyond the 600-nat guard, left to the serial chain): max |x - long double| serial glibc %.3g, serial table-driven %.3g, closed form %.3g; closed form vs serial table-driven %.3g\n", N, nguard, dg, df, dc, dcf);
    return 0;
}
