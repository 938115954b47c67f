// Shared device helpers for the gfx950 decoding kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/poreover_hip.h"

#define PO_WAVE 64
#define PO_A 4  // alphabet "ACGT"

#define PO_NEG_INF (-__builtin_inf())

// PO_EMU: the CPU lane-by-lane emulation used by tools/simt_emu (test infrastructure, never part of the shipped
// library): the few inline-assembly helpers below get a plain C++ body there.
#ifdef PO_EMU
#define PO_FMA_PLAIN 1
#define PO_NO_SLOAD 1
#endif
// v_max_f64 / v_min_f64 written out: the builtins add a canonicalisation of each operand in IEEE mode
__device__ __forceinline__ double po_vmax(double a, double b) {
#ifdef PO_EMU
    return fmax(a, b);
#else
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
__device__ __forceinline__ double po_vmin(double a, double b) {
#ifdef PO_EMU
    return fmin(a, b);
#else
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}

// Log.h:9-15 log_(): -inf for x <= 0 and for NaN
__device__ __forceinline__ double po_log_(double x) { return (x > 0) ? log(x) : PO_NEG_INF; }

// Log.h:17-23 logaddexp(): max + log_(1 + exp(min - max)); the (-inf, -inf) case yields -inf
// through log_(NaN), exactly as the reference does.
__device__ __forceinline__ double po_lae(double x1, double x2) {
    const bool ge = (x1 >= x2);
    const double hi = ge ? x1 : x2;
    const double d = ge ? (x2 - x1) : (x1 - x2);
    return hi + po_log_(1.0 + exp(d));
}

// ---- logaddexp policies ----------------------------------------------------------------------
// PoLaeOcml: Log.h's formula on the device math library (generic exp + log: ~100 f64 instructions).
// PoLaeFast: the same formula e = exp(d), z = 1 + e, log(z), with both functions specialised to the
// only ranges that occur (d <= 0, z in [1, 2]) and table-driven: ~40 f64 instructions and two LDS
// lookups.  Worst-case absolute error against a long-double reference is the same as glibc's
// log(1 + exp(d)) (1.6e-16); it differs from glibc by more than 1.2e-16 in 0.012 % of random samples
// (scripts/gen_lae_tables.py generates the tables; /tests pin parity of the decoded strings).
#include "po_lae_tables.h"
struct PoLaeTables {
    double exp_t[64][2];
    double log_t[65][3];
};
__device__ const double po_exp_t_dev[64][2] = {
#define PO_T(j) {PO_EXP_T[j][0], PO_EXP_T[j][1]}
    PO_T(0), PO_T(1), PO_T(2), PO_T(3), PO_T(4), PO_T(5), PO_T(6), PO_T(7), PO_T(8), PO_T(9), PO_T(10), PO_T(11), PO_T(12),
    PO_T(13), PO_T(14), PO_T(15), PO_T(16), PO_T(17), PO_T(18), PO_T(19), PO_T(20), PO_T(21), PO_T(22), PO_T(23), PO_T(24),
    PO_T(25), PO_T(26), PO_T(27), PO_T(28), PO_T(29), PO_T(30), PO_T(31), PO_T(32), PO_T(33), PO_T(34), PO_T(35), PO_T(36),
    PO_T(37), PO_T(38), PO_T(39), PO_T(40), PO_T(41), PO_T(42), PO_T(43), PO_T(44), PO_T(45), PO_T(46), PO_T(47), PO_T(48),
    PO_T(49), PO_T(50), PO_T(51), PO_T(52), PO_T(53), PO_T(54), PO_T(55), PO_T(56), PO_T(57), PO_T(58), PO_T(59), PO_T(60),
    PO_T(61), PO_T(62), PO_T(63)
#undef PO_T
};
__device__ const double po_log_t_dev[65][3] = {
#define PO_T(i) {PO_LOG_T[i][0], PO_LOG_T[i][1], PO_LOG_T[i][2]}
    PO_T(0), PO_T(1), PO_T(2), PO_T(3), PO_T(4), PO_T(5), PO_T(6), PO_T(7), PO_T(8), PO_T(9), PO_T(10), PO_T(11), PO_T(12),
    PO_T(13), PO_T(14), PO_T(15), PO_T(16), PO_T(17), PO_T(18), PO_T(19), PO_T(20), PO_T(21), PO_T(22), PO_T(23), PO_T(24),
    PO_T(25), PO_T(26), PO_T(27), PO_T(28), PO_T(29), PO_T(30), PO_T(31), PO_T(32), PO_T(33), PO_T(34), PO_T(35), PO_T(36),
    PO_T(37), PO_T(38), PO_T(39), PO_T(40), PO_T(41), PO_T(42), PO_T(43), PO_T(44), PO_T(45), PO_T(46), PO_T(47), PO_T(48),
    PO_T(49), PO_T(50), PO_T(51), PO_T(52), PO_T(53), PO_T(54), PO_T(55), PO_T(56), PO_T(57), PO_T(58), PO_T(59), PO_T(60),
    PO_T(61), PO_T(62), PO_T(63), PO_T(64)
#undef PO_T
};
// cooperative copy of the tables into a workgroup's LDS (call once, then barrier)
__device__ __forceinline__ void po_lae_tables_load(PoLaeTables* t, int tid, int nthr) {
    for (int i = tid; i < 64 * 2; i += nthr) (&t->exp_t[0][0])[i] = (&po_exp_t_dev[0][0])[i];
    for (int i = tid; i < 65 * 3; i += nthr) (&t->log_t[0][0])[i] = (&po_log_t_dev[0][0])[i];
}
struct PoLaeOcml {
    __device__ __forceinline__ double operator()(double x1, double x2) const { return po_lae(x1, x2); }
};
// v_fma_f64 with the constant addend in an SGPR pair and a free destination.  hipcc selects the two-address
// v_fmac_f64 for fma(x, p, c) and then copies the (loop-invariant) constant c into the destination first: one
// v_mov_b64 per polynomial step in the hottest loop.  Same instruction, same result bits.  (With the constant in
// a VGPR instead, register pressure costs a wave per SIMD in beam2d_kernel: -22 % at W = 10.)
#ifndef PO_FMA_PLAIN
__device__ __forceinline__ double po_fma_c(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
#else   // A/B switch: let the compiler pick the instruction
__device__ __forceinline__ double po_fma_c(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif
struct PoLaeFast {
    const PoLaeTables* t;
    // f(d) = log(1 + exp(d)), d <= 0.  d = NaN (from (-inf) - (-inf)) and d = -inf give 0: the caller adds it to the
    // larger operand, which is -inf in the first case — Log.h's log_(NaN) = -inf comes out without a test.
    // Integer steps instead of rint / cvt / ldexp (the loop this sits in is bound by VALU issue): k = rint(d * 64/ln2)
    // is read off the low word of d * 64/ln2 + 1.5 * 2^52; 2^m is added into the exponent field (m >= -58: no
    // subnormals); the interval of z in [1, 2] is its top six mantissa bits, rounded.  Same result bits as the
    // rint / ldexp formulation on 2e7 random and structured arguments (scripts/check_lae.c).
    // (PO_LAE_TRIM, with PO_LAE_BRANCHLESS: d0 is x1 - x2 with either sign and the clamp takes -|d0| as its operand — the
    //  source modifiers are free, and lo - hi == -|x1 - x2| bit for bit: a - b and b - a round to exact negatives)
    __device__ __forceinline__ double f(double d0) const {
#if defined(PO_LAE_BRANCHLESS) && defined(PO_LAE_TRIM)
#ifdef PO_EMU
        const double d = fmax(-fabs(d0), -40.5);
#else
        double d;
        asm("v_max_f64 %0, -|%1|, %2" : "=v"(d) : "v"(d0), "s"(-40.5));
#endif
#elif defined(PO_LAE_BRANCHLESS)
        // every lane computes; the small arguments are CLAMPED instead of tested: below -40, exp(d) < 2^-57 and 1 + e == 1
        // exactly, and that is as true of exp(-40.5) as of 0 — one v_max_f64 (which also turns d = NaN, from (-inf) - (-inf),
        // and -inf into -40.5: IEEE maxNum) instead of a compare and two selects; the table index is in range for any argument
#ifdef PO_EMU
        const double d = fmax(d0, -40.5);
#else
        double d;
        asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(d0), "s"(-40.5));
#endif
#else
        const double d = d0;
#endif
        double e = 0.0;
#ifdef PO_LAE_BRANCHLESS
        {
#else
        if (d > -40.0) {  // below: exp(d) < 2^-57, 1 + e == 1
#endif
            const double tm = __builtin_fma(d, PO_64_LN2, 0x1.8p52);
            const double kf = tm - 0x1.8p52;
            const int k = __double2loint(tm);
            const int j = k & 63;
#if defined(PO_LAE_EARLY_TABLE)
            // The table entry is asked for as soon as its index exists and nothing is scheduled across that point: left
            // alone, the scheduler sinks the LDS read below the polynomial and a lone wave waits a full LDS round trip
            // per logaddexp in its dependent chain (same instructions, same result bits).
            const double th = t->exp_t[j][0], tl = t->exp_t[j][1];
            __builtin_amdgcn_sched_barrier(0);
#endif
            double r = __builtin_fma(-kf, PO_LN2_64_HI, d);
            r = __builtin_fma(-kf, PO_LN2_64_LO, r);
            double p = po_fma_c(r, 1.0 / 720, 1.0 / 120);
            p = po_fma_c(r, p, 1.0 / 24);
            p = po_fma_c(r, p, 1.0 / 6);
            p = po_fma_c(r, p, 0.5);
            p = __builtin_fma(r * r, p, r);
#if !defined(PO_LAE_EARLY_TABLE)
            const double th = t->exp_t[j][0], tl = t->exp_t[j][1];
#endif
            const double x = th + __builtin_fma(th, p, tl);
#ifdef PO_LAE_TRIM   // (the same integer, mod 2^32, in two instructions instead of three: v_and, v_lshl_add)
#ifdef PO_EMU
            e = __hiloint2double(__double2hiint(x) + (int)((unsigned)(k & ~63) << 14), __double2loint(x));
#else
            {
                int xh;   // (written out: the compiler turns the expression back into shift, mask and add)
                asm("v_lshl_add_u32 %0, %1, 14, %2" : "=v"(xh) : "v"(k & ~63), "v"(__double2hiint(x)));
                e = __hiloint2double(xh, __double2loint(x));
            }
#endif
#else
            e = __hiloint2double(__double2hiint(x) + ((k >> 6) << 20), __double2loint(x));
#endif
        }
        const double z = 1.0 + e;
        const unsigned i = ((unsigned)__double2hiint(z) - 0x3FF00000u + 0x2000u) >> 14;
        const double* lt = (const double*)((const char*)&t->log_t[0][0] + __umul24(i, 24u));
        const double rc = lt[0], lh = lt[1], ll = lt[2];
        const double w = __builtin_fma(z, rc, -1.0);
        double q = po_fma_c(w, 1.0 / 7, -1.0 / 6);
        q = po_fma_c(w, q, 1.0 / 5);
        q = po_fma_c(w, q, -1.0 / 4);
        q = po_fma_c(w, q, 1.0 / 3);
        const double s = w * w;
        double u = __builtin_fma(s * w, q, ll);
        u = __builtin_fma(-0.5, s, u);
        return lh + (w + u);
    }
    __device__ __forceinline__ double operator()(double x1, double x2) const {
#if defined(PO_LAE_BRANCHLESS) && defined(PO_LAE_TRIM)
        const double hi = po_vmax(x1, x2);
        return hi + f(x1 - x2);
#else
        const double hi = po_vmax(x1, x2), lo = po_vmin(x1, x2);
        return hi + f(lo - hi);
#endif
    }
    // The two halves of f() on their own, for the closed-form chains (po_beam2d_reg.hip: a new element's window as ONE
    // exp per (element, time), a prefix sum in the probability domain and ONE log — not the reference's rounding, inside
    // its documented tolerance: DESIGN.md §3.3, §5).  Same tables, same polynomials.
    // ex(d) = exp(d) for d <= 0: exactly 0 below -700 (2^m is added into the exponent field: no subnormals) and for NaN
    // (-inf - -inf: a chain without a single finite term).
    __device__ __forceinline__ double ex(double d0) const {
        const double d = po_vmax(d0, -700.0);   // (maxNum: NaN -> -700; zeroed below)
        const double tm = __builtin_fma(d, PO_64_LN2, 0x1.8p52);
        const double kf = tm - 0x1.8p52;
        const int k = __double2loint(tm);
        const int j = k & 63;
        const double th = t->exp_t[j][0], tl = t->exp_t[j][1];
        double r = __builtin_fma(-kf, PO_LN2_64_HI, d);   // (|k| < 2^17: k * hi is exact, scripts/gen_lae_tables.py)
        r = __builtin_fma(-kf, PO_LN2_64_LO, r);
        double p = po_fma_c(r, 1.0 / 720, 1.0 / 120);
        p = po_fma_c(r, p, 1.0 / 24);
        p = po_fma_c(r, p, 1.0 / 6);
        p = po_fma_c(r, p, 0.5);
        p = __builtin_fma(r * r, p, r);
        const double x = th + __builtin_fma(th, p, tl);
        const double e = __hiloint2double(__double2hiint(x) + (int)((unsigned)(k & ~63) << 14), __double2loint(x));
        return (d0 >= -700.0) ? e : 0.0;
    }
    // lg(S) = log(S) for a finite S >= 0 that is 0 or normal (a sum of ex() values): -inf at 0.  S = 2^k z, z in [1, 2):
    // k ln2 (hi + lo, k * hi exact) + the table-driven log of z.
    __device__ __forceinline__ double lg(double S) const {
        const int hw = __double2hiint(S);
        const unsigned mant = (unsigned)hw & 0xFFFFFu;
        const unsigned i = (mant + 0x2000u) >> 14;
        const double z = __hiloint2double((int)(mant | 0x3FF00000u), __double2loint(S));
        const double* lt = (const double*)((const char*)&t->log_t[0][0] + __umul24(i, 24u));
        const double rc = lt[0], lh = lt[1], ll = lt[2];
        const double kd = (double)((hw >> 20) - 1023);
        const double w = __builtin_fma(z, rc, -1.0);
        double q = po_fma_c(w, 1.0 / 7, -1.0 / 6);
        q = po_fma_c(w, q, 1.0 / 5);
        q = po_fma_c(w, q, -1.0 / 4);
        q = po_fma_c(w, q, 1.0 / 3);
        const double s = w * w;
        double u = __builtin_fma(s * w, q, ll);
        u = __builtin_fma(-0.5, s, u);
        const double hi = __builtin_fma(kd, 64.0 * PO_LN2_64_HI, lh);
        const double lo = __builtin_fma(kd, 64.0 * PO_LN2_64_LO, w + u);
        return (S > 0.0) ? hi + lo : PO_NEG_INF;
    }
};

// PoLaePoly: table-free variant of the same formula (no LDS lookups): exp by k*ln2 range reduction and a
// degree-12 polynomial, log1p(e) as 2*atanh(e / (2 + e)) with an odd series.  More f64 instructions, no
// LDS round trips.
struct PoLaePoly {
    __device__ __forceinline__ double f(double d) const {
        double res = 0.0;
        if (d > -40.0) {
            const double kf = rint(d * 1.4426950408889634);
            const int k = (int)kf;
            double r = __builtin_fma(-kf, 6.93147180369123816490e-01, d);
            r = __builtin_fma(-kf, 1.90821492927058770002e-10, r);
            double p = 1.0 / 479001600;
            p = __builtin_fma(r, p, 1.0 / 39916800);
            p = __builtin_fma(r, p, 1.0 / 3628800);
            p = __builtin_fma(r, p, 1.0 / 362880);
            p = __builtin_fma(r, p, 1.0 / 40320);
            p = __builtin_fma(r, p, 1.0 / 5040);
            p = __builtin_fma(r, p, 1.0 / 720);
            p = __builtin_fma(r, p, 1.0 / 120);
            p = __builtin_fma(r, p, 1.0 / 24);
            p = __builtin_fma(r, p, 1.0 / 6);
            p = __builtin_fma(r, p, 0.5);
            p = __builtin_fma(r * r, p, r);
            const double e = ldexp(1.0 + p, k);
            const double s = e / (2.0 + e);
            const double s2 = s * s;
            double q = 1.0 / 33;
            q = __builtin_fma(s2, q, 1.0 / 31);
            q = __builtin_fma(s2, q, 1.0 / 29);
            q = __builtin_fma(s2, q, 1.0 / 27);
            q = __builtin_fma(s2, q, 1.0 / 25);
            q = __builtin_fma(s2, q, 1.0 / 23);
            q = __builtin_fma(s2, q, 1.0 / 21);
            q = __builtin_fma(s2, q, 1.0 / 19);
            q = __builtin_fma(s2, q, 1.0 / 17);
            q = __builtin_fma(s2, q, 1.0 / 15);
            q = __builtin_fma(s2, q, 1.0 / 13);
            q = __builtin_fma(s2, q, 1.0 / 11);
            q = __builtin_fma(s2, q, 1.0 / 9);
            q = __builtin_fma(s2, q, 1.0 / 7);
            q = __builtin_fma(s2, q, 1.0 / 5);
            q = __builtin_fma(s2, q, 1.0 / 3);
            const double s3 = s2 * s;
            res = 2.0 * __builtin_fma(s3, q, s);
        }
        return (d == d) ? res : PO_NEG_INF;
    }
    __device__ __forceinline__ double operator()(double x1, double x2) const {
        const bool ge = (x1 >= x2);
        const double hi = ge ? x1 : x2;
        const double d = ge ? (x2 - x1) : (x1 - x2);
        return hi + f(d);
    }
};

__device__ __forceinline__ int po_lane() { return threadIdx.x & (PO_WAVE - 1); }

// The value of lane ^ 32 (the other half of the wave): v_permlane32_swap_b32 (gfx950) exchanges the upper half of one
// register with the lower half of another inside the VALU — two of them and two selects per double, instead of two
// ds_bpermute_b32 round trips through the LDS crossbar.
__device__ __forceinline__ int po_xor32_i(int x, bool upper) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return (int)(upper ? r[0] : r[1]);
}
// x + (x of lane ^ 32): after swapping the upper half of one copy with the lower half of another, every lane holds its own
// value in one of the two and the other half's in the other — their sum needs no select (a + b == b + a bit for bit)
__device__ __forceinline__ double po_sum32(double x) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double po_xor32(double x, bool upper) {   // upper: this lane is one of 32..63
    return __hiloint2double(po_xor32_i(__double2hiint(x), upper), po_xor32_i(__double2loint(x), upper));
}

// The maximum of x over the wave, in every lane (all 64 lanes must be executing): four DPP exchanges inside the rows of 16
// (quad xor 1, quad xor 2, half-row mirror, row mirror — a maximum does not care which partner it sees), then the four
// rows through v_readlane.  No LDS round trips: ~ 25 instructions.
__device__ __forceinline__ double po_wave_max(double x) {
#define PO_DPP_D(ctrl) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), (ctrl), 0xf, 0xf, false), \
                                        __builtin_amdgcn_update_dpp(0, __double2loint(x), (ctrl), 0xf, 0xf, false))
    x = po_vmax(x, PO_DPP_D(0xB1));    // quad_perm [1,0,3,2]
    x = po_vmax(x, PO_DPP_D(0x4E));    // quad_perm [2,3,0,1]
    x = po_vmax(x, PO_DPP_D(0x141));   // row_half_mirror
    x = po_vmax(x, PO_DPP_D(0x140));   // row_mirror
#undef PO_DPP_D
    auto rl = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l)); };
    return po_vmax(po_vmax(rl(0), rl(16)), po_vmax(rl(32), rl(48)));
}

// "These registers are final HERE."  gfx9 counts a wave's vector loads AND stores on one counter (vmcnt) and they complete in
// order, so `s_waitcnt vmcnt(0)` in front of the first use of a loaded value also waits for every store the wave has in flight.
// The compiler places that wait where the value is first used — after the join of a RARE branch that loaded it, that is on the
// common path too, where nothing was loaded and a full store queue is drained for nothing (po_beam2d_reg.hip's table build: a
// round trip of ~ 2.5 us per beam change at full load, 11 % of the kernel: round 6).  Called at the end of the rare branch, this
// makes the branch itself wait: past the join nothing is pending.
__device__ __forceinline__ void po_settle(int& a, int& b) {
#ifndef PO_EMU
    asm volatile("" : "+v"(a), "+v"(b));
#else
    (void)a; (void)b;
#endif
}
__device__ __forceinline__ void po_settle(int& a) {
#ifndef PO_EMU
    asm volatile("" : "+v"(a));
#else
    (void)a;
#endif
}
__device__ __forceinline__ void po_settle(double& a) {
#ifndef PO_EMU
    asm volatile("" : "+v"(a));
#else
    (void)a;
#endif
}

// The lane number as a value the compiler cannot carry from somewhere else: an address built from it is computed where it is
// used (two or three VALU operations).  Hoisted out of the walk loop instead, such an address is one more kernel-lifetime
// register — the ones that get SPILLED — and on gfx9 a scratch reload is a vector load: the wait in front of its first use is
// `s_waitcnt vmcnt(0)`, which drains the wave's store queue (po_settle above).  Used where a spilled LDS address was reloaded
// once per main step (po_beam2d_reg.hip: sm.pf0).
__device__ __forceinline__ int po_lane_here() {
    int l = (int)threadIdx.x;
#ifndef PO_EMU
    asm volatile("" : "+v"(l));
#endif
    return l & (PO_WAVE - 1);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global access (s_waitcnt vmcnt(0)); inside a dependent loop that turns each
// fire-and-forget global store into a full round trip.  Use this one when only LDS data is
// exchanged across the barrier and global visibility is established later by __syncthreads().
__device__ __forceinline__ void po_lds_barrier() {
#ifdef PO_EMU
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// LDS hand-over inside a ONE-WAVE workgroup: a wave's LDS (and vector memory) operations are performed in program
// order, so only the compiler needs a fence — __syncthreads() would also drain every outstanding global store
// (s_waitcnt vmcnt(0)), a full round trip each time in a loop that writes tree nodes as it goes.
__device__ __forceinline__ void po_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Candidate ordering used by every prune: higher score first; exact ties by node creation
// order (ascending id).  The reference's tie order is heap-address order (Beam.h:96-107).
__device__ __forceinline__ bool po_better(double sa, int ia, double sb, int ib) {
    return (sa > sb) || (!(sb > sa) && ia < ib);
}

// ---- exact score ties: what Beam::prune leaves (Beam.h:93-108) -----------------------------------------------
// The reference sorts the candidate POINTERS (nodes are allocated one after the other and never freed during a
// search, so pointer order is creation order, i.e. node id order), removes duplicates, and then calls
// std::partial_sort(first, first + W, last, greater-by-score) — or std::sort when there are at most W candidates.
// With distinct scores every correct algorithm returns the same beam; with EXACT ties (quantised uint8 traces,
// candidates that are all -inf) the beam is whatever libstdc++'s heap-select / sort_heap / introsort leave.  The
// functions below restate those algorithms (bits/stl_heap.h, bits/stl_algo.h) on an index array `o` (slots in
// node-id order) with score(slot) given by a functor; one lane runs them, and only when a tie touches the beam.
template <class S>
__device__ inline void po_stl_push_heap(int* o, int hole, int top, int v, const S& sc) {
    int parent = (hole - 1) / 2;
    while (hole > top && sc(o[parent]) > sc(v)) { o[hole] = o[parent]; hole = parent; parent = (hole - 1) / 2; }
    o[hole] = v;
}
template <class S>
__device__ inline void po_stl_adjust_heap(int* o, int hole, int len, int v, const S& sc) {
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (sc(o[second]) > sc(o[second - 1])) second--;
        o[hole] = o[second]; hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        o[hole] = o[second - 1]; hole = second - 1;
    }
    po_stl_push_heap(o, hole, top, v, sc);
}
template <class S>
__device__ inline void po_stl_partial_sort(int* o, int mid, int n, const S& sc) {
    if (mid == 0) return;
    if (mid >= 2)
        for (int parent = (mid - 2) / 2;; --parent) { const int v = o[parent]; po_stl_adjust_heap(o, parent, mid, v, sc); if (parent == 0) break; }
    for (int i = mid; i < n; ++i)
        if (sc(o[i]) > sc(o[0])) { const int v = o[i]; o[i] = o[0]; po_stl_adjust_heap(o, 0, mid, v, sc); }
    for (int last = mid; last > 1;) { --last; const int v = o[last]; o[last] = o[0]; po_stl_adjust_heap(o, 0, last, v, sc); }
}
template <class S>
__device__ inline void po_stl_unguarded_linear_insert(int* o, int last, const S& sc) {
    const int v = o[last];
    int next = last - 1;
    while (sc(v) > sc(o[next])) { o[last] = o[next]; last = next; --next; }
    o[last] = v;
}
template <class S>
__device__ inline void po_stl_insertion_sort(int* o, int first, int last, const S& sc) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (sc(o[i]) > sc(o[first])) { const int v = o[i]; for (int k = i; k > first; --k) o[k] = o[k - 1]; o[first] = v; }
        else po_stl_unguarded_linear_insert(o, i, sc);
    }
}
// (stk: 48 ints for the explicit stack when MAXN > 16 — a caller in a hot kernel hands LDS: a private array lives in
//  scratch memory; the overload without it keeps its own)
template <int MAXN, class S>
__device__ inline void po_stl_sort(int* o, int n, const S& sc, int* stk) {   // std::sort of n <= MAXN <= 64 elements
    if (n == 0) return;
    if (MAXN <= 16) {   // (introsort leaves ranges of <= 16 elements to the final insertion sort)
        po_stl_insertion_sort(o, 0, n, sc);
        return;
    }
    int lg = 0;
    for (int k = n; k > 1; k >>= 1) ++lg;
    // __introsort_loop without recursion: an explicit stack of (first, last, depth)
    int sp = 0;
    stk[0] = 0; stk[1] = n; stk[2] = 2 * lg; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stk[3 * sp], last = stk[3 * sp + 1], depth = stk[3 * sp + 2];
        while (last - first > 16) {
            if (depth == 0) { po_stl_partial_sort(o + first, last - first, last - first, sc); break; }
            --depth;
            const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
            auto sw = [&](int x, int y) { const int t = o[x]; o[x] = o[y]; o[y] = t; };
            if (sc(o[a]) > sc(o[b])) { if (sc(o[b]) > sc(o[c])) sw(first, b); else if (sc(o[a]) > sc(o[c])) sw(first, c); else sw(first, a); }
            else if (sc(o[a]) > sc(o[c])) sw(first, a);
            else if (sc(o[b]) > sc(o[c])) sw(first, c);
            else sw(first, b);
            int lo = first + 1, hi = last;
            for (;;) {
                while (sc(o[lo]) > sc(o[first])) ++lo;
                --hi;
                while (sc(o[first]) > sc(o[hi])) --hi;
                if (!(lo < hi)) break;
                sw(lo, hi);
                ++lo;
            }
            // the reference recurses on [lo, last) first and then loops on [first, lo): order does not matter for
            // the result (the two ranges are disjoint)
            if (sp < 16) { stk[3 * sp] = lo; stk[3 * sp + 1] = last; stk[3 * sp + 2] = depth; ++sp; }
            last = lo;
        }
    }
    if (n > 16) { po_stl_insertion_sort(o, 0, 16, sc); for (int i = 16; i != n; ++i) po_stl_unguarded_linear_insert(o, i, sc); }
    else po_stl_insertion_sort(o, 0, n, sc);
}
// the W best of the n candidates o[0..n) (slots in node-id order) exactly as Beam::prune orders them, in o[0..min(W, n))
template <int MAXN, class S>
__device__ inline void po_stl_sort(int* o, int n, const S& sc) {
    int stk[(MAXN > 16) ? 48 : 1];
    po_stl_sort<MAXN>(o, n, sc, stk);
}
template <int WMAX, class S>
__device__ inline void po_stl_prune(int* o, int n, int W, const S& sc, int* stk) {   // W <= WMAX; stk: 48 ints (see po_stl_sort)
    if (n > W) po_stl_partial_sort(o, W, n, sc);
    else po_stl_sort<WMAX>(o, n, sc, stk);
}
template <int WMAX, class S>
__device__ inline void po_stl_prune(int* o, int n, int W, const S& sc) {
    if (n > W) po_stl_partial_sort(o, W, n, sc);
    else po_stl_sort<WMAX>(o, n, sc);
}

// node arena entry: parent id and last symbol packed as (parent << 3) | last  (last <= 4)
__device__ __forceinline__ int po_pack_node(int parent, int last) { return (parent << 3) | last; }
__device__ __forceinline__ int po_node_parent(int packed) { return packed >> 3; }
__device__ __forceinline__ int po_node_last(int packed) { return packed & 7; }

// ---- tree-model recurrences shared by the 1-D and pair kernels ------------------------------
// root values at time tm1 (tree constructors PrefixTree.h:467-476, :541-546, :641-647)
template <int MODEL>
__device__ __forceinline__ void root_values(int tm1, double blank_cum, double* out) {
    if (MODEL == PO_MODEL_CTC) {
        out[0] = (tm1 < 0) ? 0.0 : blank_cum;
    } else if (MODEL == PO_MODEL_MERGE) {
        out[0] = (tm1 < 0) ? 0.0 : PO_NEG_INF;
        out[1] = (tm1 < 0) ? 0.0 : PO_NEG_INF;
        out[2] = PO_NEG_INF;
    } else {
        const double h = log(0.5);
        out[0] = (tm1 < 0) ? 0.0 : PO_NEG_INF;
        out[1] = (tm1 < 0) ? h : PO_NEG_INF;
        out[2] = (tm1 < 0) ? h : PO_NEG_INF;
    }
}


// update_prob for one node at one time (PrefixTree.h:478-488,518-531 ctc; :649-663,690-704 merge
// repeats; :548-574,600-632 flip-flop).  sp = own values at t-1, pp = parent's values at t-1,
// ya = y[t][last]; yb = y[t][blank] (CTC models) or y[t][last + A] (flip-flop);
// same = (parent->last == last); first = (parent->depth == 0 && t == 0).
template <int MODEL, class LAE = PoLaeOcml>
__device__ __forceinline__ void po_update(const double* sp, const double* pp, double ya, double yb, bool same,
                                          bool first, double* out, const LAE& po_lae = LAE()) {
    if (MODEL == PO_MODEL_CTC) {
        out[0] = po_lae(pp[0] + ya, sp[0] + yb);
    } else if (MODEL == PO_MODEL_MERGE) {
        // (ONE logaddexp whatever `same` says: the operand is selected, not the branch — lanes of one wave differ in it,
        //  and two calls behind a divergent branch run one after the other)
        const double gap = sp[0] + yb;
        const double x = same ? pp[1] : pp[0];
        const double lg = po_lae(x + ya, sp[2] + ya);
        const double ng = first ? ya : lg;
        out[0] = po_lae(gap, ng);
        out[1] = gap;
        out[2] = ng;
    } else {
        // Three logaddexp per update, not four: the reference's `else` case (parent->last != last) computes
        // emit_flip = logaddexp(p.flip, p.flop) + y and flop = logaddexp(-inf, stay_flop) — and logaddexp(-inf, x) is x for
        // every x (x + log(1 + exp(-inf)) = x + 0; (-inf, -inf) gives -inf either way: Log.h:9-23, and the same holds for the
        // table-driven form) — while the `same` case needs a logaddexp for flop and none for the emission.  One call serves
        // both, its operands selected per lane.
        const double stay_flip = sp[1] + ya;
        const double stay_flop = sp[2] + yb;
        const bool fs = first || same;
        const double eo_fs = first ? yb : pp[1] + yb;
        const double l2 = po_lae(fs ? eo_fs : pp[1], fs ? stay_flop : pp[2]);
        const double ef = first ? ya : (same ? pp[2] + ya : l2 + ya);
        const double flip = po_lae(ef, stay_flip);
        const double flop = fs ? l2 : stay_flop;
        out[0] = po_lae(flip, flop);
        out[1] = flip;
        out[2] = flop;
    }
}

// hipMemsetAsync's replacement on the decode path: the runtime's fill kernel comes in 256-thread workgroups, which find no
// room next to a device full of one-wave pair beam workgroups (the pipelined job: a 256-byte queue reset waited 12 ms for
// the launch before it to drain).  One-wave workgroups fit wherever one of those has left a slot.  bytes: a multiple of 4.
// (only in the sources that define PO_WANT_ZERO_KERNEL before including this header)
#if !defined(PO_EMU) && defined(PO_WANT_ZERO_KERNEL)
static __global__ __launch_bounds__(64) void po_zero_kernel(unsigned* p, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 64) p[i] = 0u;
}
static inline hipError_t po_zero_async(void* p, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) != 0 || ((size_t)p & 3) != 0 || bytes > ((size_t)4 << 20)) return hipMemsetAsync(p, 0, bytes, stream);   // (large: the runtime's)
    const size_t nwords = bytes >> 2;
    const size_t nb_ = (nwords + 63) / 64;
    const unsigned blocks = (unsigned)(nb_ < 4096 ? nb_ : 4096);
    hipLaunchKernelGGL(po_zero_kernel, dim3(blocks), dim3(64), 0, stream, (unsigned*)p, nwords);
    return hipGetLastError();
}
#endif

