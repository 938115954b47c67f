/* TEST INFRASTRUCTURE ONLY — see po_oracle.h.  Parity status: PINNED (see header).
 *
 * Plain-C restatement of the reference algorithm.  Every function cites the reference
 * file:line (relative to /root/reference/poreover/) whose behaviour it restates.  The
 * restatement is literal about the reference's quirks (sparse-in-time alpha maps, growing
 * beam vectors, uninitialised-band UB is turned into PO_E_ENVELOPE, ...).  The only
 * deliberate deviations, both documented in DESIGN.md:
 *   - Beam<T,F>::prune (decoding/Beam.h:93-108) breaks exact score ties by heap address;
 *     here ties break by node creation order (ascending arena index).
 *   - inputs on which the reference reads out of bounds / uninitialised memory / never
 *     terminates return a PO_E_* code instead.
 *
 * Storage: the reference keeps per-node std::unordered_map<int,double> (absent == -inf on
 * read, PrefixTree.h:55-61).  Here each node owns a dense window of doubles that grows on
 * demand and is initialised to -inf, which is read-for-read equivalent.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fPIC -shared (see Makefile); no -ffast-math.
 */
#include "po_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-INFINITY)

/* ------------------------------------------------------------------ Log.h:9-23 */
static double log_(double x) { return (x > 0) ? log(x) : NEG_INF; }

double oracle_logaddexp(double x1, double x2) {
    if (x1 >= x2) return x1 + log_(1 + exp(x2 - x1));
    return x2 + log_(1 + exp(x1 - x2));
}
#define LAE oracle_logaddexp

/* numpy's np.logaddexp (npy_logaddexp), used by the pure-python twins in prefix_search.py */
static double np_logaddexp(double x, double y) {
    if (x == y) return x + 0.693147180559945309417232121458176568; /* LOGE2 */
    double tmp = x - y;
    if (tmp > 0) return x + log1p(exp(-tmp));
    if (tmp <= 0) return y + log1p(exp(tmp));
    return tmp; /* NaN */
}

/* =====================================================================================
 * Prefix tree with per-node, per-read time-indexed values          (PrefixTree.h:17-706)
 * ===================================================================================== */
typedef struct {
    int t0, n, cap;
    double* v; /* K channels interleaved: v[(t - t0) * K + ch] */
} track_t;

typedef struct {
    int parent, last, depth, first_child;
    track_t tr[2];
    int last_t[2], max_t[2];
    double last_prob[2], max_prob[2];
} node_t;

typedef struct {
    int model, K, A, C;
    const double* y[2];
    int T[2];
    node_t* nodes;
    int n, cap;
    int oom;
} tree_t;

static double track_get(const track_t* tr, int K, int ch, int t) {
    if (t < tr->t0 || t >= tr->t0 + tr->n) return NEG_INF;
    return tr->v[(size_t)(t - tr->t0) * K + ch];
}

static int track_set(track_t* tr, int K, int t, const double* vals) {
    if (tr->n == 0 && tr->v == NULL) {
        tr->cap = 16;
        tr->v = (double*)malloc(sizeof(double) * K * tr->cap);
        if (!tr->v) return -1;
        tr->t0 = t;
        tr->n = 0;
    }
    if (t < tr->t0 || t >= tr->t0 + tr->cap) {
        int lo = t < tr->t0 ? t : tr->t0;
        int hi = (t + 1 > tr->t0 + tr->n) ? t + 1 : tr->t0 + tr->n;
        int need = hi - lo;
        int ncap = need * 2 + 16;
        int nlo = (t < tr->t0) ? lo - need / 2 : lo; /* slack on the growing side */
        double* nv = (double*)malloc(sizeof(double) * K * ncap);
        if (!nv) return -1;
        for (size_t i = 0; i < (size_t)K * ncap; ++i) nv[i] = NEG_INF;
        memcpy(nv + (size_t)(tr->t0 - nlo) * K, tr->v, sizeof(double) * K * tr->n);
        free(tr->v);
        tr->v = nv;
        tr->n += tr->t0 - nlo;
        tr->t0 = nlo;
        tr->cap = ncap;
    }
    while (tr->t0 + tr->n <= t) { /* extend with absent (-inf) entries */
        for (int k = 0; k < K; ++k) tr->v[(size_t)tr->n * K + k] = NEG_INF;
        tr->n++;
    }
    for (int k = 0; k < K; ++k) tr->v[(size_t)(t - tr->t0) * K + k] = vals[k];
    return 0;
}

static int tree_new_node(tree_t* tr, int parent, int last) {
    if (tr->n == tr->cap) {
        int nc = tr->cap ? tr->cap * 2 : 1024;
        node_t* nn = (node_t*)realloc(tr->nodes, sizeof(node_t) * nc);
        if (!nn) { tr->oom = 1; return -1; }
        tr->nodes = nn;
        tr->cap = nc;
    }
    node_t* nd = &tr->nodes[tr->n];
    memset(nd, 0, sizeof(*nd));
    nd->parent = parent;
    nd->last = last;
    nd->depth = parent >= 0 ? tr->nodes[parent].depth + 1 : 0;
    nd->first_child = -1;
    nd->max_prob[0] = nd->max_prob[1] = NEG_INF; /* PrefixTree.h:81,206,344 */
    return tr->n++;
}

/* tree constructors: PrefixTree.h:467-476 (ctc 1D), :499-516 (ctc 2D), :541-546/:585-598
 * (flip-flop), :641-647/:674-688 (merge-repeats).  dims = 1 or 2. */
static int tree_init(tree_t* tr, int model, int A, int C, const double* y1, int T1, const double* y2,
                     int T2) {
    memset(tr, 0, sizeof(*tr));
    tr->model = model;
    tr->A = A;
    tr->C = C;
    tr->K = (model == PO_MODEL_CTC) ? 1 : 3;
    tr->y[0] = y1; tr->T[0] = T1;
    tr->y[1] = y2; tr->T[1] = T2;
    int dims = y2 ? 2 : 1;
    int root = tree_new_node(tr, -1, A); /* root->last = gap_char / flipflop_size = |alphabet| */
    if (root < 0) return -1;
    for (int d = 0; d < dims; ++d) {
        track_t* t = &tr->nodes[root].tr[d];
        if (model == PO_MODEL_CTC) {
            double z = 0;
            if (track_set(t, 1, -1, &z)) return -1;
            double blank_sum = 0;
            for (int i = 0; i < tr->T[d]; ++i) {
                blank_sum += tr->y[d][(size_t)i * C + A];
                if (track_set(t, 1, i, &blank_sum)) return -1;
            }
        } else if (model == PO_MODEL_MERGE) {
            double v[3] = {0, 0, NEG_INF}; /* total, gap, no_gap */
            if (track_set(t, 3, -1, v)) return -1;
        } else {
            double h = log(0.5);
            double v[3] = {0, h, h}; /* total, flip, flop */
            if (track_set(t, 3, -1, v)) return -1;
        }
    }
    return 0;
}

static void tree_free(tree_t* tr) {
    for (int i = 0; i < tr->n; ++i) {
        free(tr->nodes[i].tr[0].v);
        free(tr->nodes[i].tr[1].v);
    }
    free(tr->nodes);
}

/* PrefixTree<TNode>::expand, PrefixTree.h:439-446 */
static int tree_expand(tree_t* tr, int n) {
    if (tr->nodes[n].first_child < 0) {
        int fc = -1;
        for (int i = 0; i < tr->A; ++i) {
            int c = tree_new_node(tr, n, i);
            if (c < 0) return -1;
            if (i == 0) fc = c;
        }
        tr->nodes[n].first_child = fc;
    }
    return tr->nodes[n].first_child;
}

/* set_probability: PrefixTree.h:69-72 (1D), :129-137 (2D ctc), :269-278, :407-416 */
static void node_set(tree_t* tr, int n, int d, int t, const double* vals) {
    node_t* nd = &tr->nodes[n];
    if (track_set(&nd->tr[d], tr->K, t, vals)) tr->oom = 1;
    nd->last_t[d] = t;
    nd->last_prob[d] = vals[0];
    if (vals[0] > nd->max_prob[d]) {
        nd->max_t[d] = t;
        nd->max_prob[d] = vals[0];
    }
}

#define P(n, d, ch, t) track_get(&tr->nodes[(n)].tr[(d)], tr->K, (ch), (t))

/* update_prob: PrefixTree.h:478-488,518-531 (ctc); :649-663,690-704 (merge repeats);
 * :548-574,600-632 (flip-flop).  d = read index. */
static void tree_update(tree_t* tr, int n, int d, int t) {
    const node_t* nd = &tr->nodes[n];
    const int par = nd->parent, last = nd->last, A = tr->A;
    const double* yt = tr->y[d] + (size_t)t * tr->C;
    if (tr->model == PO_MODEL_CTC) {
        double emit_state = P(par, d, 0, t - 1) + yt[last];
        double stay_state = P(n, d, 0, t - 1) + yt[A];
        double v = LAE(emit_state, stay_state);
#ifdef PO_ORACLE_TRACE_NODE
        if (n == PO_ORACLE_TRACE_NODE) printf("V %d %d %d %.17g %.17g %.17g\n", n, d, t, v, P(par, d, 0, t - 1), P(n, d, 0, t - 1));
#endif
        node_set(tr, n, d, t, &v);
    } else if (tr->model == PO_MODEL_MERGE) {
        double gap_prob = P(n, d, 0, t - 1) + yt[A];
        double no_gap;
        if (tr->nodes[par].depth == 0 && t == 0) {
            no_gap = yt[last];
        } else if (tr->nodes[par].last == last) {
            no_gap = LAE(P(par, d, 1, t - 1) + yt[last], P(n, d, 2, t - 1) + yt[last]);
        } else {
            no_gap = LAE(P(par, d, 0, t - 1) + yt[last], P(n, d, 2, t - 1) + yt[last]);
        }
        double v[3] = {LAE(gap_prob, no_gap), gap_prob, no_gap};
        node_set(tr, n, d, t, v);
    } else {
        double stay_flip = P(n, d, 1, t - 1) + yt[last];
        double stay_flop = P(n, d, 2, t - 1) + yt[last + A];
        double emit_flip, emit_flop;
        if (tr->nodes[par].depth == 0 && t == 0) {
            emit_flip = yt[last];
            emit_flop = yt[last + A];
        } else if (tr->nodes[par].last == last) {
            emit_flip = P(par, d, 2, t - 1) + yt[last];
            emit_flop = P(par, d, 1, t - 1) + yt[last + A];
        } else {
            emit_flip = LAE(P(par, d, 1, t - 1), P(par, d, 2, t - 1)) + yt[last];
            emit_flop = NEG_INF;
        }
        double flip = LAE(emit_flip, stay_flip);
        double flop = LAE(emit_flop, stay_flop);
        double v[3] = {LAE(flip, flop), flip, flop};
        node_set(tr, n, d, t, v);
    }
}

static void node_reset_max(tree_t* tr, int n) { /* PrefixTree.h:115-118 */
    tr->nodes[n].max_prob[0] = NEG_INF;
    tr->nodes[n].max_prob[1] = NEG_INF;
}

/* get_label, PrefixTree.h:449-457; the root's '\0' is stripped as decoding_cpp.pyx:101 does */
static int tree_label(const tree_t* tr, int n, const char* alphabet, char* out, int cap) {
    int len = tr->nodes[n].depth;
    if (len + 1 > cap) return PO_E_CAP;
    out[len] = '\0';
    for (int i = len - 1, p = n; i >= 0; --i, p = tr->nodes[p].parent) out[i] = alphabet[tr->nodes[p].last];
    return len;
}

/* ------------------------------------------------------------------ Beam.h:76-114 */
typedef struct { int* el; int n, cap, width; } beam_t;

static int beam_push(beam_t* b, int x) {
    if (b->n == b->cap) {
        int nc = b->cap ? b->cap * 2 : 64;
        int* ne = (int*)realloc(b->el, sizeof(int) * nc);
        if (!ne) return -1;
        b->el = ne;
        b->cap = nc;
    }
    b->el[b->n++] = x;
    return 0;
}

enum { SCORE_LAST_1D, SCORE_ROW, SCORE_ROW_COL, SCORE_GRID };
static double node_score(const tree_t* tr, int n, int kind) {
    const node_t* nd = &tr->nodes[n];
    switch (kind) {
        case SCORE_LAST_1D: return nd->last_prob[0];                   /* node_greater, 1D   */
        case SCORE_ROW: return nd->last_prob[0] + nd->max_prob[1];     /* node_greater_max   */
        case SCORE_ROW_COL: return nd->max_prob[0] + nd->max_prob[1];  /* ..._max_sym        */
        default: return nd->last_prob[0] + nd->last_prob[1];           /* node_greater, 2D   */
    }
}

typedef struct { double s; int id; } scored_t;
static int scored_cmp(const void* a, const void* b) {
    const scored_t* x = (const scored_t*)a; const scored_t* y = (const scored_t*)b;
    if (x->s > y->s) return -1;
    if (y->s > x->s) return 1;
    return (x->id > y->id) - (x->id < y->id); /* documented tie rule: creation order */
}
static int int_cmp(const void* a, const void* b) {
    int x = *(const int*)a, y = *(const int*)b;
    return (x > y) - (x < y);
}

/* ---- libstdc++'s std::partial_sort / std::sort, restated (bits/stl_heap.h, bits/stl_algo.h of GCC's libstdc++):
 * with exact score ties the result of Beam::prune is whatever these algorithms leave, starting from the
 * pointer-sorted vector.  Nodes are allocated one after the other and never freed during a search, so pointer order
 * is taken to be creation order (ids).  comp(a, b) = score(a) > score(b), as the reference's comparators. */
static int g_tie_stl = 1;   /* 1: what libstdc++ leaves (the reference); 0: score desc, then creation order (round-1 rule, kept for A/B) */
void oracle_set_tie_rule(int stl) { g_tie_stl = stl; }
#define SCOMP(a, b) ((a).s > (b).s)
static void stl_push_heap(scored_t* f, long hole, long top, scored_t v) {
    long parent = (hole - 1) / 2;
    while (hole > top && SCOMP(f[parent], v)) { f[hole] = f[parent]; hole = parent; parent = (hole - 1) / 2; }
    f[hole] = v;
}
static void stl_adjust_heap(scored_t* f, long hole, long len, scored_t v) {
    const long top = hole;
    long second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (SCOMP(f[second], f[second - 1])) second--;
        f[hole] = f[second]; hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        f[hole] = f[second - 1]; hole = second - 1;
    }
    stl_push_heap(f, hole, top, v);
}
static void stl_make_heap(scored_t* f, long len) {
    if (len < 2) return;
    long parent = (len - 2) / 2;
    for (;;) { scored_t v = f[parent]; stl_adjust_heap(f, parent, len, v); if (parent == 0) return; parent--; }
}
static void stl_pop_heap(scored_t* f, long len, scored_t* result) { scored_t v = *result; *result = f[0]; stl_adjust_heap(f, 0, len, v); }
static void stl_partial_sort(scored_t* f, long mid, long n) {
    if (mid == 0) return;
    stl_make_heap(f, mid);
    for (long i = mid; i < n; ++i) if (SCOMP(f[i], f[0])) stl_pop_heap(f, mid, &f[i]);
    long last = mid;
    while (last > 1) { --last; stl_pop_heap(f, last, &f[last]); }
}
static void stl_unguarded_linear_insert(scored_t* f, long last) {
    scored_t v = f[last];
    long next = last - 1;
    while (SCOMP(v, f[next])) { f[last] = f[next]; last = next; --next; }
    f[last] = v;
}
static void stl_insertion_sort(scored_t* f, long first, long last) {
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (SCOMP(f[i], f[first])) { scored_t v = f[i]; memmove(&f[first + 1], &f[first], sizeof(scored_t) * (size_t)(i - first)); f[first] = v; }
        else stl_unguarded_linear_insert(f, i);
    }
}
static void stl_swap(scored_t* a, scored_t* b) { scored_t t = *a; *a = *b; *b = t; }
static void stl_introsort_loop(scored_t* f, long first, long last, long depth) {
    while (last - first > 16) {
        if (depth == 0) { stl_partial_sort(f + first, last - first, last - first); return; }
        --depth;
        const long mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
        if (SCOMP(f[a], f[b])) { if (SCOMP(f[b], f[c])) stl_swap(&f[first], &f[b]); else if (SCOMP(f[a], f[c])) stl_swap(&f[first], &f[c]); else stl_swap(&f[first], &f[a]); }
        else if (SCOMP(f[a], f[c])) stl_swap(&f[first], &f[a]);
        else if (SCOMP(f[b], f[c])) stl_swap(&f[first], &f[c]);
        else stl_swap(&f[first], &f[b]);
        long lo = first + 1, hi = last;
        for (;;) {
            while (SCOMP(f[lo], f[first])) ++lo;
            --hi;
            while (SCOMP(f[first], f[hi])) --hi;
            if (!(lo < hi)) break;
            stl_swap(&f[lo], &f[hi]);
            ++lo;
        }
        stl_introsort_loop(f, lo, last, depth);
        last = lo;
    }
}
static void stl_sort(scored_t* f, long n) {
    if (n == 0) return;
    long lg = 0;
    for (long k = n; k > 1; k >>= 1) ++lg;
    stl_introsort_loop(f, 0, n, 2 * lg);
    if (n > 16) { stl_insertion_sort(f, 0, 16); for (long i = 16; i != n; ++i) stl_unguarded_linear_insert(f, i); }
    else stl_insertion_sort(f, 0, n);
}

/* Beam::prune, Beam.h:93-108: dedupe by identity, keep the top `width` by comparator */
static int beam_prune(beam_t* b, const tree_t* tr, int kind) {
    qsort(b->el, b->n, sizeof(int), int_cmp);
    int m = 0;
    for (int i = 0; i < b->n; ++i)
        if (i == 0 || b->el[i] != b->el[i - 1]) b->el[m++] = b->el[i];
    scored_t* sc = (scored_t*)malloc(sizeof(scored_t) * (m ? m : 1));
    if (!sc) return -1;
    for (int i = 0; i < m; ++i) { sc[i].id = b->el[i]; sc[i].s = node_score(tr, b->el[i], kind); }
    if (g_tie_stl) {
        if (m > b->width) stl_partial_sort(sc, b->width, m);
        else stl_sort(sc, m);
    } else {
        qsort(sc, m, sizeof(scored_t), scored_cmp);
    }
    if (m > b->width) m = b->width;
    for (int i = 0; i < m; ++i) b->el[i] = sc[i].id;
    b->n = m;
    free(sc);
    return 0;
}

/* =====================================================================================
 * 1-D beam search                                                   (BeamSearch.h:18-58)
 * ===================================================================================== */
static int check_model(int model, int A, int C) {
    if (A < 1) return 0;
    if (model == PO_MODEL_CTC || model == PO_MODEL_MERGE) return C == A + 1;
    if (model == PO_MODEL_FLIPFLOP) return C == 2 * A;
    return 0;
}

int oracle_beam_search_1d(const double* y, int T, int C, const char* alphabet, int W, int model,
                          char* out, int cap) {
    int A = (int)strlen(alphabet);
    if (!check_model(model, A, C) || T < 1 || W < 1) return PO_E_ARG;
    tree_t tr;
    beam_t beam = {0};
    beam.width = W;
    int rc = PO_E_NOMEM;
    if (tree_init(&tr, model, A, C, y, T, NULL, 0)) goto done;
    {
        int fc = tree_expand(&tr, 0);
        if (fc < 0) goto done;
        for (int i = 0; i < A; ++i) {
            tree_update(&tr, fc + i, 0, 0);
            if (beam_push(&beam, fc + i)) goto done;
        }
    }
    for (int t = 1; t < T; ++t) {
        int beam_size = beam.n;
        for (int b = 0; b < beam_size; ++b) {
            int node = beam.el[b];
            tree_update(&tr, node, 0, t);
            int fc = tree_expand(&tr, node);
            if (fc < 0) goto done;
            for (int i = 0; i < A; ++i) {
                tree_update(&tr, fc + i, 0, t);
                if (beam_push(&beam, fc + i)) goto done;
            }
        }
        if (beam_prune(&beam, &tr, SCORE_LAST_1D)) goto done;
        if (tr.oom) goto done;
    }
    rc = tree_label(&tr, beam.el[0], alphabet, out, cap);
done:
    tree_free(&tr);
    free(beam.el);
    return rc;
}

/* forward_ / forward, PrefixTree.h:710-759: log P(label | y) by chaining update_prob.
 * Unknown label characters map to index 0 (std::unordered_map default-insert, :724). */
double oracle_forward(const double* y, int T, int C, const char* label, const char* alphabet, int model) {
    int A = (int)strlen(alphabet);
    if (!check_model(model, A, C) || T < 1) return NAN;
    tree_t tr;
    double res = NAN;
    if (tree_init(&tr, model, A, C, y, T, NULL, 0)) { tree_free(&tr); return NAN; }
    int cur = 0;
    for (const char* s = label; *s; ++s) {
        const char* p = strchr(alphabet, *s);
        int li = p ? (int)(p - alphabet) : 0;
        cur = tree_new_node(&tr, cur, li); /* add_child, not expand */
        if (cur < 0) goto done;
        for (int t = 0; t < T; ++t) tree_update(&tr, cur, 0, t);
    }
    if (cur == 0) { /* root->last_probability() == probability.at(max_t = 0) */
        res = (model == PO_MODEL_CTC) ? track_get(&tr.nodes[0].tr[0], 1, 0, 0) : NAN;
    } else {
        res = tr.nodes[cur].last_prob[0];
    }
done:
    tree_free(&tr);
    return res;
}

/* =====================================================================================
 * 2-D beam searches                         (BeamSearch.h:110-397, BeamSearch2.h:33-184)
 * ===================================================================================== */
static int beam2d_seed(tree_t* tr, beam_t* beam) {
    int fc = tree_expand(tr, 0);
    if (fc < 0) return -1;
    for (int i = 0; i < tr->A; ++i) {
        tree_update(tr, fc + i, 0, 0);
        tree_update(tr, fc + i, 1, 0);
        if (beam_push(beam, fc + i)) return -1;
    }
    return 0;
}

/* beam_search_2d_by_row: with envelope BeamSearch.h:110-172, without :175-260 */
static int beam2d_row(tree_t* tr, beam_t* beam, const int* env, int U, int V) {
    const int A = tr->A, W = beam->width;
    for (int u = env ? 0 : 1; u < U; ++u) {
        int rs = env ? env[2 * u] : 0, re = env ? env[2 * u + 1] : V;
        if (rs < 0 || re > V) return PO_E_ENVELOPE; /* reference would read y2 out of bounds */
        for (int b = 0; b < W; ++b) { /* loops over a GROWING vector, :132-144 */
            if (b >= beam->n) return PO_E_ENVELOPE;
            int node = beam->el[b];
            tree_update(tr, node, 0, u);
            int fc = tree_expand(tr, node);
            if (fc < 0) return PO_E_NOMEM;
            for (int i = 0; i < A; ++i) {
                tree_update(tr, fc + i, 0, u);
                if (beam_push(beam, fc + i)) return PO_E_NOMEM;
            }
        }
        for (int v = rs; v < re; ++v)
            for (int b = 0; b < beam->n; ++b) {
                if (v == rs) node_reset_max(tr, beam->el[b]);
                tree_update(tr, beam->el[b], 1, v);
            }
        if (beam_prune(beam, tr, SCORE_ROW)) return PO_E_NOMEM;
        if (tr->oom) return PO_E_NOMEM;
    }
    return 0;
}

#ifdef PO_ORACLE_STATS
/* Workload statistics of the row_col walk (scripts/rowcol_stats.py builds a scratch library with this flag; the
 * shipped oracle does not carry it).  Indices: see scripts/rowcol_stats.py. */
long long g_rc_stats[64];
long long* oracle_rowcol_stats(void) { return g_rc_stats; }
static int rc_in(const int* el, int n, int x) { for (int i = 0; i < n; ++i) if (el[i] == x) return 1; return 0; }
#endif

/* beam_search_2d_by_row_col, BeamSearch.h:262-397 */
static int beam2d_row_col(tree_t* tr, beam_t* beam, const int* env, int U, int V) {
    const int A = tr->A, W = beam->width;
    int* envt = (int*)malloc(sizeof(int) * 2 * (V > 0 ? V : 1));
    if (!envt) return PO_E_NOMEM;
    int rc = 0;
    for (int v = 0; v < V; ++v) envt[2 * v] = envt[2 * v + 1] = -1;
    for (int u = 0; u < U; ++u)
        for (int x = env[2 * u]; x < env[2 * u + 1]; ++x) {
            if (x < 0 || x >= V) { rc = PO_E_ENVELOPE; goto done; } /* :276-282 overflow */
            if (envt[2 * x] < 0) { envt[2 * x] = u; envt[2 * x + 1] = u + 1; }
            else envt[2 * x + 1]++;
        }
    int u = 0, v = 0;
#ifdef PO_ORACLE_STATS
    int we0 = -1, we1 = -1, pmaxw = 0;
#endif
    while (u <= U - 1 && v <= V - 1) {
        int ers = env[2 * u], ere = env[2 * u + 1];
        int ecs = envt[2 * v], ece = envt[2 * v + 1];
        int row_start = 0, row_end = 0, col_start = 0, col_end = 0, row_ok = 0, col_ok = 0;
        if (v >= ers && v < ere) {
            row_start = v; row_end = ere; row_ok = 1;
        } else if (v < ers) { /* catch-up along read 1, :314-322 */
            int nb = W < beam->n ? W : beam->n; /* reference indexes b < beam_width (UB if larger) */
#ifdef PO_ORACLE_STATS
            g_rc_stats[1]++; if (v >= we1) g_rc_stats[2]++;
#endif
            for (int b = 0; b < nb; ++b) tree_update(tr, beam->el[b], 1, v);
            v++;
            continue;
        }
        if (u >= ecs && u < ece) {
            col_start = u; col_end = ece; col_ok = 1;
        } else if (u < ecs) { /* catch-up along read 0, :328-336 */
            int nb = W < beam->n ? W : beam->n;
#ifdef PO_ORACLE_STATS
            g_rc_stats[1]++; if (u >= we0) g_rc_stats[2]++;
#endif
            for (int b = 0; b < nb; ++b) tree_update(tr, beam->el[b], 0, u);
            u++;
            continue;
        }
        if (!row_ok || !col_ok) { rc = PO_E_ENVELOPE; goto done; } /* uninitialised bounds, :309 */

        int beam_size = beam->n;
#ifdef PO_ORACLE_STATS
        int old[64], nold = beam_size < 64 ? beam_size : 64;
        for (int b = 0; b < nold; ++b) old[b] = beam->el[b];
        {
            int l0 = col_end - col_start, l1 = row_end - row_start;
            g_rc_stats[0]++; g_rc_stats[3] += l0; g_rc_stats[4] += l1;
            if (l0 > pmaxw) pmaxw = l0;
            if (l1 > pmaxw) pmaxw = l1;
            if (col_end < we0) g_rc_stats[5]++;
            if (row_end < we1) g_rc_stats[5]++;
            /* new times this step if everything were incremental */
            g_rc_stats[6] += (col_end > we0 ? col_end - (we0 > col_start ? we0 : col_start) : 0);
            g_rc_stats[7] += (row_end > we1 ? row_end - (we1 > row_start ? we1 : row_start) : 0);
            we0 = col_end; we1 = row_end;
        }
#endif
        for (int b = 0; b < beam_size; ++b) {
            int node = beam->el[b];
            tree_update(tr, node, 0, col_start);
            tree_update(tr, node, 1, row_start);
            node_reset_max(tr, node);
            int fc = tree_expand(tr, node);
            if (fc < 0) { rc = PO_E_NOMEM; goto done; }
            for (int i = 0; i < A; ++i) {
                node_reset_max(tr, fc + i);
                tree_update(tr, fc + i, 0, col_start);
                tree_update(tr, fc + i, 1, row_start);
                if (beam_push(beam, fc + i)) { rc = PO_E_NOMEM; goto done; }
            }
        }
        for (int u_ = col_start; u_ < col_end; ++u_)
            for (int b = 0; b < beam->n; ++b) tree_update(tr, beam->el[b], 0, u_);
        for (int v_ = row_start; v_ < row_end; ++v_)
            for (int b = 0; b < beam->n; ++b) tree_update(tr, beam->el[b], 1, v_);
#ifdef PO_ORACLE_TRACE   /* scratch builds only (scripts/trace_rowcol.py): every candidate's score before the prune */
        for (int b = 0; b < beam->n; ++b)
            printf("T %d %d %d %.17g\n", u, v, beam->el[b], node_score(tr, beam->el[b], SCORE_ROW_COL));
#endif
        if (beam_prune(beam, tr, SCORE_ROW_COL) || tr->oom) { rc = PO_E_NOMEM; goto done; }
#ifdef PO_ORACLE_STATS
        {
            int nn = beam->n, same = (nn == nold), sameset = (nn == nold), enter = 0;
            for (int b = 0; b < nn; ++b) {
                if (b >= nold || beam->el[b] != old[b]) same = 0;
                if (!rc_in(old, nold, beam->el[b])) {
                    sameset = 0; enter++;
                    const node_t* nd = &tr->nodes[beam->el[b]];
                    if (nd->first_child >= 0) {
                        g_rc_stats[12]++;   /* re-entry: node already has children */
                        const node_t* c = &tr->nodes[nd->first_child];
                        if (c->last_t[0] >= u || c->last_t[1] >= v) g_rc_stats[13]++;   /* ... whose stale values the next step can read */
                    }
                }
            }
            if (same) g_rc_stats[8]++; else if (sameset) g_rc_stats[9]++; else g_rc_stats[10]++;
            g_rc_stats[11] += enter;
            if (enter > 1) g_rc_stats[20]++;
            for (int b = 0; b < nn; ++b) {   /* frozen parents */
                const node_t* nd = &tr->nodes[beam->el[b]];
                int p = nd->parent;
                if (p <= 0) continue;
                int gp = tr->nodes[p].parent;
                if (rc_in(beam->el, nn, p) || (gp >= 0 && rc_in(beam->el, nn, gp))) continue;
                g_rc_stats[14]++;
                const node_t* pn = &tr->nodes[p];
                if (pn->last_t[0] > nd->last_t[0] || pn->last_t[1] > nd->last_t[1]) g_rc_stats[15]++;  /* claim violated */
                if (pn->last_t[0] == nd->last_t[0] || pn->last_t[1] == nd->last_t[1]) g_rc_stats[16]++; /* frozen value at done-1 exists */
            }
            if (nn == W) {   /* exact tie at the beam boundary or inside it */
                int tie = 0;
                for (int b = 0; b + 1 < nn; ++b)
                    if (node_score(tr, beam->el[b], SCORE_ROW_COL) == node_score(tr, beam->el[b + 1], SCORE_ROW_COL)) tie = 1;
                if (tie) g_rc_stats[17]++;
            }
        }
#endif
        v++;
        u++;
    }
#ifdef PO_ORACLE_STATS
    g_rc_stats[18]++;                       /* pairs */
    if (pmaxw > g_rc_stats[19]) g_rc_stats[19] = pmaxw;
    if (pmaxw <= 14) g_rc_stats[21]++;
    else if (pmaxw <= 30) g_rc_stats[22]++;
    else if (pmaxw <= 62) g_rc_stats[23]++;
    else g_rc_stats[24]++;
    g_rc_stats[25] += tr->n;                /* nodes created */
#endif
done:
    free(envt);
    return rc;
}

/* beam_search_2d_grid: without envelope BeamSearch2.h:33-119, with :121-184.
 * One beam per visited cell; predecessor = cell (u-1, v-1), or the seed beam when that
 * cell is outside the stored band (SparseMatrix default value, SparseMatrix.h:51-57). */
static int beam2d_grid(tree_t* tr, int W, const int* env, int U, int V, int* top_out) {
    const int A = tr->A;
    beam_t seed = {0};
    seed.width = W;
    int rc = 0;
    size_t* row_off = (size_t*)calloc((size_t)U + 1, sizeof(size_t));
    beam_t* cells = NULL;
    if (!row_off) return PO_E_NOMEM;
    if (beam2d_seed(tr, &seed)) { rc = PO_E_NOMEM; goto done; }
    for (int u = 0; u < U; ++u) {
        int rs = env ? env[2 * u] : 0, re = env ? env[2 * u + 1] : V;
        if (rs < 0 || re > V) { rc = PO_E_ENVELOPE; goto done; }
        /* stored columns: [rs, re] inclusive with envelope (push_row), [0, V) without */
        int width = env ? (re - rs + 1) : V;
        if (width < 0) width = 0;
        row_off[u + 1] = row_off[u] + (size_t)width;
    }
    cells = (beam_t*)calloc(row_off[U] ? row_off[U] : 1, sizeof(beam_t));
    if (!cells) { rc = PO_E_NOMEM; goto done; }
#define CELL(u_, v_) (&cells[row_off[(u_)] + (size_t)((v_) - (env ? env[2 * (u_)] : 0))])
#define IN_BAND(u_, v_) ((u_) >= 0 && (u_) < U && (v_) >= (env ? env[2 * (u_)] : 0) && \
                         (v_) <= (env ? env[2 * (u_) + 1] : V - 1))
    for (int u = 0; u < U; ++u) {
        int rs = env ? env[2 * u] : 0, re = env ? env[2 * u + 1] : V;
        for (int v = rs; v < re; ++v) {
            beam_t* cur = CELL(u, v);
            cur->width = W;
            const beam_t* prev = &seed;
            if (u > 0 && v > 0 && IN_BAND(u - 1, v - 1)) {
                const beam_t* c = CELL(u - 1, v - 1);
                /* a stored-but-never-visited cell (v == re of its row) still holds the default */
                if (c->width != 0) prev = c;
            }
            for (int k = 0; k < prev->n; ++k) {
                int node = prev->el[k];
                tree_update(tr, node, 0, u);
                tree_update(tr, node, 1, v);
                if (beam_push(cur, node)) { rc = PO_E_NOMEM; goto done; }
                int fc = tree_expand(tr, node);
                if (fc < 0) { rc = PO_E_NOMEM; goto done; }
                for (int i = 0; i < A; ++i) {
                    tree_update(tr, fc + i, 0, u);
                    tree_update(tr, fc + i, 1, v);
                    if (beam_push(cur, fc + i)) { rc = PO_E_NOMEM; goto done; }
                }
            }
            if (beam_prune(cur, tr, SCORE_GRID) || tr->oom) { rc = PO_E_NOMEM; goto done; }
        }
    }
    {
        const beam_t* last = &seed;
        if (IN_BAND(U - 1, V - 1) && CELL(U - 1, V - 1)->width != 0) last = CELL(U - 1, V - 1);
        if (last->n < 1) { rc = PO_E_ENVELOPE; goto done; }
        if (getenv("PO_ORACLE_DEBUG"))
            for (int k = 0; k < last->n; ++k)
                fprintf(stderr, "grid final[%d] node=%d score=%g\n", k, last->el[k], node_score(tr, last->el[k], SCORE_GRID));
        *top_out = last->el[0];
    }
done:
    if (cells) for (size_t i = 0; i < row_off[U]; ++i) free(cells[i].el);
    free(cells);
    free(row_off);
    free(seed.el);
    return rc;
}

int oracle_beam_search_2d(const double* y1, int U, const double* y2, int V, int C,
                          const char* alphabet, const int* env, int W, int model, int method,
                          char* out, int cap) {
    int A = (int)strlen(alphabet);
    if (!check_model(model, A, C) || U < 1 || V < 1 || W < 1) return PO_E_ARG;
    /* dispatcher BeamSearch.h:441-458: without an envelope only "row" and grid exist */
    if (!env && method == PO_METHOD_ROW_COL) method = PO_METHOD_GRID;
    tree_t tr;
    beam_t beam = {0};
    beam.width = W;
    int rc = PO_E_NOMEM;
    if (tree_init(&tr, model, A, C, y1, U, y2, V)) goto done;
    if (method == PO_METHOD_GRID) {
        int top = -1;
        rc = beam2d_grid(&tr, W, env, U, V, &top);
        if (rc == 0) rc = tree_label(&tr, top, alphabet, out, cap);
        goto done;
    }
    if (beam2d_seed(&tr, &beam)) goto done;
    if (method == PO_METHOD_ROW) rc = beam2d_row(&tr, &beam, env, U, V);
    else if (method == PO_METHOD_ROW_COL) rc = beam2d_row_col(&tr, &beam, env, U, V);
    else rc = PO_E_ARG;
    if (rc == 0) rc = tree_label(&tr, beam.el[0], alphabet, out, cap);
done:
    tree_free(&tr);
    free(beam.el);
    return rc;
}

/* =====================================================================================
 * argmax / Viterbi decoders                                       (transducer.py:27-106)
 * ===================================================================================== */
int oracle_argmax_path(const double* y, int T, int C, int* path) { /* np.argmax: first max */
    for (int t = 0; t < T; ++t) {
        const double* r = y + (size_t)t * C;
        int best = 0;
        for (int c = 1; c < C; ++c)
            if (r[c] > r[best]) best = c;
        path[t] = best;
    }
    return T;
}

int oracle_viterbi_decode(const double* y, int T, int C, int kind, const char* alphabet, int* path,
                          char* seq, int cap) {
    const char* dna = alphabet ? alphabet : "ACGT";
    int n = 0;
    if (T < 1) return PO_E_ARG;
    if (kind == PO_KIND_POREOVER || kind == PO_KIND_BONITO) {
        /* poreover.viterbi_decode == argmax_decode (:72-73); bonito adds groupby (:83-89) */
        if (C < 2 || (int)strlen(dna) != C - 1) return PO_E_ARG;
        oracle_argmax_path(y, T, C, path);
        for (int t = 0; t < T; ++t) {
            if (kind == PO_KIND_BONITO && t > 0 && path[t] == path[t - 1]) continue;
            if (path[t] == C - 1) continue; /* alphabet[-1] == '' */
            if (n + 1 >= cap) return PO_E_CAP;
            seq[n++] = dna[path[t]];
        }
        seq[n] = '\0';
        return n;
    }
    if (kind != PO_KIND_FLIPFLOP || C != 8) return PO_E_ARG;
    /* transducer.viterbi_decode :35-59 with the flip-flop transition matrix :94-103.
     * NOTE the 0/1 "transition" is ADDED to log-probabilities (:44), as the reference does. */
    static const char ff[] = "ACGTacgt";
    const int S = 8, A = 4;
    double v[8], nv[8];
    int8_t* ptr = (int8_t*)malloc((size_t)T * S);
    if (!ptr) return PO_E_NOMEM;
    for (int j = 0; j < S; ++j) v[j] = y[j];
    for (int t = 1; t < T; ++t) {
        const double* r = y + (size_t)t * S;
        for (int j = 0; j < S; ++j) {
            int bi = 0;
            double bv = 0;
            for (int i = 0; i < S; ++i) {
                double tr_ij = (j < A) ? 1.0 : ((i % A) == (j - A) ? 1.0 : 0.0);
                double cand = tr_ij + v[i];
                if (i == 0 || cand > bv) { bv = cand; bi = i; }
            }
            ptr[(size_t)t * S + j] = (int8_t)bi;
            nv[j] = r[j] + bv;
        }
        memcpy(v, nv, sizeof(v));
    }
    int best = 0;
    for (int j = 1; j < S; ++j)
        if (v[j] > v[best]) best = j;
    path[T - 1] = best;
    for (int i = T - 2; i >= 0; --i) path[i] = ptr[(size_t)(i + 1) * S + path[i + 1]];
    free(ptr);
    for (int t = 0; t < T; ++t) { /* remove_repeated(...).upper(), :4-9,55 */
        if (t > 0 && path[t] == path[t - 1]) continue;
        if (n + 1 >= cap) return PO_E_CAP;
        seq[n++] = ff[path[t] % A];
    }
    seq[n] = '\0';
    return n;
}

/* =====================================================================================
 * Viterbi acceptors                          (Forward.h:14-121, decoding_cy.pyx:60-123)
 * ===================================================================================== */
typedef struct { int start, end; double* v; int* p; } srow_t; /* SparseRow, inclusive [start,end] */

int oracle_viterbi_acceptor(const double* y, int T, int C, int band, const char* label,
                            const char* alphabet, int* path) {
    const int A = (int)strlen(alphabet), gap = A, L = (int)strlen(label);
    if (C != A + 1 || T < 1 || L < 1 || band < 0) return PO_E_ARG;
    int rc = T;
    int* li = (int*)malloc(sizeof(int) * L);
    int nrows = L + 2; /* two pre-pushed rows + one per label position, Forward.h:42-45,62 */
    srow_t* rows = (srow_t*)calloc(nrows, sizeof(srow_t));
    if (!li || !rows) { free(li); free(rows); return PO_E_NOMEM; }
    for (int l = 0; l < L; ++l) { const char* p = strchr(alphabet, label[l]); li[l] = p ? (int)(p - alphabet) : 0; }
    int nr = 0;
#define PUSH_ROW(s_, e_) do { int w_ = (e_) - (s_) + 1; if (w_ < 0) w_ = 0; \
        rows[nr].start = (s_); rows[nr].end = (e_); \
        rows[nr].v = (double*)malloc(sizeof(double) * (w_ ? w_ : 1)); \
        rows[nr].p = (int*)malloc(sizeof(int) * (w_ ? w_ : 1)); \
        if (!rows[nr].v || !rows[nr].p) { rc = PO_E_NOMEM; nr++; goto done; } \
        for (int q_ = 0; q_ < w_; ++q_) { rows[nr].v[q_] = NEG_INF; rows[nr].p[q_] = 0; } nr++; } while (0)
#define VGET(r_, c_) (((r_) >= 0 && (r_) < nr && (c_) >= rows[(r_)].start && (c_) <= rows[(r_)].end) ? rows[(r_)].v[(c_) - rows[(r_)].start] : NEG_INF)
#define PGET(r_, c_) (((r_) >= 0 && (r_) < nr && (c_) >= rows[(r_)].start && (c_) <= rows[(r_)].end) ? rows[(r_)].p[(c_) - rows[(r_)].start] : 0)
#define VSET(r_, c_, x_) do { if ((r_) >= 0 && (r_) < nr && (c_) >= rows[(r_)].start && (c_) <= rows[(r_)].end) rows[(r_)].v[(c_) - rows[(r_)].start] = (x_); } while (0)
#define PSET(r_, c_, x_) do { if ((r_) >= 0 && (r_) < nr && (c_) >= rows[(r_)].start && (c_) <= rows[(r_)].end) rows[(r_)].p[(c_) - rows[(r_)].start] = (x_); } while (0)
    PUSH_ROW(0, band);
    PUSH_ROW(0, band);
    {
        double gap_prob = 0;
        for (int t = 0; t < T; ++t) { gap_prob += y[(size_t)t * C + gap]; VSET(0, t, gap_prob); PSET(0, t, 0); }
    }
    VSET(1, 0, y[li[0]]);
    PSET(0, 0, 0);
    PSET(1, 0, 1);
    for (int l = 1; l <= L; ++l) {
        int c = (int)(l * (double)T / (double)L);
        int rs = (1 > c - band) ? 1 : c - band;
        int re = (T < c + band) ? T : c + band;
        PUSH_ROW(rs, re); /* lands at row index l+1: the reference's off-by-one, :62-63 */
        for (int t = rs; t < re; ++t) {
            if (t >= l - 1) {
                double emit = y[(size_t)t * C + li[l - 1]] + VGET(l - 1, t - 1);
                double stay = y[(size_t)t * C + gap] + VGET(l, t - 1);
                if (emit >= stay) { VSET(l, t, emit); PSET(l, t, 1); }
                else { VSET(l, t, stay); PSET(l, t, 0); }
            }
        }
    }
    for (int t = 0; t < T; ++t) path[t] = gap;
    {
        int l = L, t = T - 1;
        while (l > 0) {
            if (t < 0) { rc = PO_E_DIVERGE; goto done; } /* reference loops forever here */
            if (PGET(l, t) > 0) { path[t] = li[l - 1]; l -= 1; }
            t -= 1;
        }
    }
done:
    for (int i = 0; i < nr; ++i) { free(rows[i].v); free(rows[i].p); }
    free(rows);
    free(li);
    return rc;
}

/* decoding_cy.viterbi_acceptor, decoding_cy.pyx:60-123 (dense; '>' tie rule; the band
 * expression reuses the loop variable t from the previous loop, :103, language_level=2
 * integer division for l/l_max). */
int oracle_viterbi_acceptor_cy(const double* y, int T, int C, int band, const char* label,
                               const char* alphabet, int* path) {
    const int A = (int)strlen(alphabet), gap = A, L = (int)strlen(label);
    if (C != A + 1 || T < 1 || L < 1) return PO_E_ARG;
    int* li = (int*)malloc(sizeof(int) * L);
    double* v = (double*)malloc(sizeof(double) * (size_t)(L + 1) * T);
    int8_t* ptr = (int8_t*)malloc((size_t)(L + 1) * T);
    if (!li || !v || !ptr) { free(li); free(v); free(ptr); return PO_E_NOMEM; }
    int rc = T;
    for (int l = 0; l < L; ++l) {
        const char* p = strchr(alphabet, label[l]);
        if (!p) { rc = PO_E_ARG; goto done; } /* KeyError in the reference */
        li[l] = (int)(p - alphabet);
    }
    for (size_t i = 0; i < (size_t)(L + 1) * T; ++i) { v[i] = NEG_INF; ptr[i] = (int8_t)gap; }
    int band_ = band > 0 ? band : T;
    long t = 0;
    {
        double g = 0;
        for (t = 0; t < T; ++t) { g += y[(size_t)t * C + gap]; v[t] = g; ptr[t] = (int8_t)gap; }
        t = T - 1; /* Python loop variable keeps its last value */
    }
    v[(size_t)1 * T + 0] = y[li[0]];
    ptr[0] = 1;
    for (long l = 1; l <= L; ++l) {
        long c = (l / L) * t; /* floor division of non-negative ints */
        long lo = (1 > c - band_) ? 1 : c - band_;
        long hi = (T < c + band_) ? T : c + band_;
        for (long tt = lo; tt < hi; ++tt) {
            t = tt;
            if (tt >= l) {
                double emit = y[(size_t)tt * C + li[l - 1]] + v[(size_t)(l - 1) * T + tt - 1];
                double stay = y[(size_t)tt * C + gap] + v[(size_t)l * T + tt - 1];
                if (emit > stay) { v[(size_t)l * T + tt] = emit; ptr[(size_t)l * T + tt] = 1; }
                else { v[(size_t)l * T + tt] = stay; ptr[(size_t)l * T + tt] = 0; }
            }
        }
    }
    for (int i = 0; i < T; ++i) path[i] = gap;
    {
        long l = L, tt = T - 1;
        while (l > 0) {
            if (tt < -T) { rc = PO_E_DIVERGE; goto done; }
            long idx = tt < 0 ? tt + T : tt; /* wraparound=True, :61 */
            if (ptr[(size_t)l * T + idx]) { path[idx] = li[l - 1]; l -= 1; }
            tt -= 1;
        }
    }
done:
    free(li); free(v); free(ptr);
    return rc;
}

/* =====================================================================================
 * Gamma DP                      (Gamma.h:15-98; decoding_cy.pyx:177-220; prefix_search.py:35-65)
 * ===================================================================================== */
static double pair_gamma_envelope_impl(const double* y1, const double* y2, const int* env, int U, int V, int C, double* dense) {
    /* env: U+1 rows, inclusive [start, end] (SparseMatrix.h:35-57); default -inf outside */
    size_t* off = (size_t*)malloc(sizeof(size_t) * (U + 2));
    if (!off) return NAN;
    off[0] = 0;
    for (int u = 0; u <= U; ++u) {
        int w = env[2 * u + 1] - env[2 * u] + 1;
        off[u + 1] = off[u] + (size_t)(w > 0 ? w : 0);
    }
    double* g = (double*)malloc(sizeof(double) * (off[U + 1] ? off[U + 1] : 1));
    double* ga = (double*)malloc(sizeof(double) * (off[U + 1] ? off[U + 1] : 1));
    if (!g || !ga) { free(off); free(g); free(ga); return NAN; }
    for (size_t i = 0; i < off[U + 1]; ++i) g[i] = ga[i] = NEG_INF;
#define GIN(u_, v_) ((u_) >= 0 && (u_) <= U && (v_) >= env[2 * (u_)] && (v_) <= env[2 * (u_) + 1])
#define GGET(m_, u_, v_) (GIN(u_, v_) ? (m_)[off[(u_)] + (size_t)((v_) - env[2 * (u_)])] : NEG_INF)
#define GSET(m_, u_, v_, x_) do { if (GIN(u_, v_)) (m_)[off[(u_)] + (size_t)((v_) - env[2 * (u_)])] = (x_); } while (0)
    const int b = C - 1;
    GSET(g, U, V, 0.0);
    GSET(ga, U, V, 0.0);
    for (int v = 0; v < V; ++v) {
        double s = 0.;
        for (int ve = v; ve < V; ++ve) s += y2[(size_t)ve * C + b];
        GSET(g, U, v, s);
    }
    for (int u = 0; u < U; ++u) {
        double s = 0.;
        for (int ue = u; ue < U; ++ue) s += y1[(size_t)ue * C + b];
        GSET(g, u, V, s);
    }
    for (int u = U - 1; u >= 0; --u) {
        int rs = env[2 * u], re = env[2 * u + 1] - 1;
        for (int v = re; v >= rs; --v) {
            if (v < 0 || v >= V) continue; /* reference would read y2 out of bounds */
            double gamma_eps = GGET(g, u + 1, v) + y1[(size_t)u * C + b];
            double gamma_ast_eps = GGET(ga, u, v + 1) + y2[(size_t)v * C + b];
            double total2 = 0.;
            for (int t = 0; t < C - 1; ++t) total2 += exp(y1[(size_t)u * C + t] + y2[(size_t)v * C + t]);
            double gamma_ast_ast = GGET(g, u + 1, v + 1) + log(total2);
            double x = LAE(gamma_ast_eps, gamma_ast_ast);
            GSET(ga, u, v, x);
            x = LAE(gamma_eps, GGET(ga, u, v));
            GSET(g, u, v, x);
        }
    }
    double res = GGET(g, 0, 0);
    if (dense) /* the (U+1) x (V+1) matrix with -inf outside the stored ranges (what SparseMatrix::get returns there) */
        for (int u = 0; u <= U; ++u)
            for (int v = 0; v <= V; ++v) dense[(size_t)u * (V + 1) + v] = GGET(g, u, v);
    free(off); free(g); free(ga);
    return res;
}
double oracle_pair_gamma_envelope(const double* y1, const double* y2, const int* env, int U, int V, int C) {
    return pair_gamma_envelope_impl(y1, y2, env, U, V, C, NULL);
}

static double seq_logsumexp_shift(const double* a, int n, int stride) {
    /* scipy.special.logsumexp: max-shifted (summation order differs from numpy's pairwise
     * sum; agreement is to rounding, tests use np.isclose like the reference's own) */
    double m = NEG_INF;
    for (int i = 0; i < n; ++i) if (a[(size_t)i * stride] > m) m = a[(size_t)i * stride];
    if (!isfinite(m)) m = 0;
    double s = 0;
    for (int i = 0; i < n; ++i) s += exp(a[(size_t)i * stride] - m);
    return log(s) + m;
}

int oracle_pair_gamma_dense(const double* y1, int U, const double* y2, int V, int C, int flavor,
                            double* gm) {
    const double LOG0 = flavor ? -9999.0 : NEG_INF;
    const int b = C - 1;
    const size_t W1 = (size_t)V + 1;
    double* ga = (double*)malloc(sizeof(double) * (U + 1) * W1);
    if (!ga) return PO_E_NOMEM;
    for (size_t i = 0; i < (size_t)(U + 1) * W1; ++i) gm[i] = ga[i] = LOG0;
    gm[(size_t)U * W1 + V] = 0;
    ga[(size_t)U * W1 + V] = 0;
    for (int v = 0; v < V; ++v) { double s = 0; for (int k = v; k < V; ++k) s += y2[(size_t)k * C + b]; gm[(size_t)U * W1 + v] = s; }
    for (int u = 0; u < U; ++u) { double s = 0; for (int k = u; k < U; ++k) s += y1[(size_t)k * C + b]; gm[(size_t)u * W1 + V] = s; }
    for (int u = U - 1; u >= 0; --u)
        for (int v = V - 1; v >= 0; --v) {
            double ge = gm[(size_t)(u + 1) * W1 + v] + y1[(size_t)u * C + b];
            double gae = ga[(size_t)u * W1 + v + 1] + y2[(size_t)v * C + b];
            double gaa;
            if (flavor) {
                double tot = 0;
                for (int t = 0; t < C - 1; ++t) tot += exp(y1[(size_t)u * C + t] + y2[(size_t)v * C + t]);
                gaa = gm[(size_t)(u + 1) * W1 + v + 1] + log(tot);
                ga[(size_t)u * W1 + v] = log(exp(gae) + exp(gaa));
                gm[(size_t)u * W1 + v] = log(exp(ge) + exp(ga[(size_t)u * W1 + v]));
            } else {
                double tmp[16];
                int nn = C - 1 < 16 ? C - 1 : 16;
                for (int t = 0; t < nn; ++t) tmp[t] = y1[(size_t)u * C + t] + y2[(size_t)v * C + t];
                gaa = gm[(size_t)(u + 1) * W1 + v + 1] + seq_logsumexp_shift(tmp, nn, 1);
                ga[(size_t)u * W1 + v] = np_logaddexp(gae, gaa);
                gm[(size_t)u * W1 + v] = np_logaddexp(ge, ga[(size_t)u * W1 + v]);
            }
        }
    free(ga);
    return 0;
}

/* =====================================================================================
 * Prefix search                (prefix_search.py:67-385; decoding_cy.pyx:127-156,326-347)
 * ===================================================================================== */
int oracle_forward_vec_log(int s, int i, const double* y, int T, int C, const double* previous,
                           int flavor, double* fw) {
    /* s == -1 selects the blank column (python negative index), prefix_search.py:81-96 */
    const double LOG0 = flavor ? -9999.0 : NEG_INF;
    const int sc = s < 0 ? C + s : s;
    if (i != 0 && !previous) return PO_E_ARG;
    for (int t = 0; t < T; ++t) fw[t] = LOG0;
    for (int t = 0; t < T; ++t) {
        const double* r = y + (size_t)t * C;
        if (i == 0) {
            fw[t] = (t == 0) ? r[sc] : r[C - 1] + fw[t - 1];
        } else if (t == 0) {
            if (i == 1) fw[t] = r[sc];
        } else if (flavor) {
            fw[t] = log(exp(r[C - 1] + fw[t - 1]) + exp(r[sc] + previous[t - 1]));
        } else {
            fw[t] = np_logaddexp(r[C - 1] + fw[t - 1], r[sc] + previous[t - 1]);
        }
    }
    return T;
}

/* forward_vec_no_gap_log, prefix_search.py:67-79 */
static void fw_no_gap(int len, int lastc, const double* y, int T, int C, const double* fw0, double* out) {
    for (int t = 0; t < T; ++t) {
        double prev = (t == 0) ? (len == 1 ? 0.0 : NEG_INF) : fw0[t - 1];
        out[t] = prev + y[(size_t)t * C + lastc];
    }
}

/* prefix_search_log / prefix_search_log_cy, prefix_search.py:115-238 */
int oracle_prefix_search_log(const double* y, int T, int C, int flavor, char* label, int cap, double* logp) {
    static const char dna[] = "ACGT";
    const int A = C - 1;
    if (T < 1 || A < 1 || A > 4) return PO_E_ARG;
    double* alpha_prev = (double*)malloc(sizeof(double) * T);
    double* alphas = (double*)malloc(sizeof(double) * T * A);
    double* ast = (double*)malloc(sizeof(double) * T);
    char* curr = (char*)calloc(T + 2, 1);
    char* top = (char*)calloc(T + 2, 1);
    int rc = PO_E_NOMEM;
    if (!alpha_prev || !alphas || !ast || !curr || !top) goto done;
    double top_prob = 0; /* label_prob[''] = gap_prob = np.sum(y[:,-1]) */
    for (int t = 0; t < T; ++t) top_prob += y[(size_t)t * C + A];
    int curr_len = 0, top_len = 0, level = 0;
    oracle_forward_vec_log(-1, 0, y, T, C, NULL, flavor, alpha_prev);
    for (;;) {
        level++;
        if (level > T) { rc = PO_E_DIVERGE; goto done; } /* prefix_forward index error upstream */
        int best_c = 0;
        double prefix_prob[4], best_prefix_prob = 0;
        for (int c = 0; c < A; ++c) {
            fw_no_gap(curr_len + 1, c, y, T, C, alpha_prev, ast);
            prefix_prob[c] = seq_logsumexp_shift(ast, T, 1);
            double* alpha = alphas + (size_t)c * T;
            oracle_forward_vec_log(c, level, y, T, C, alpha_prev, flavor, alpha);
            double lp = alpha[T - 1];
            if (lp > top_prob) { /* label_prob[prefix] > label_prob[top_label] */
                top_prob = lp;
                memcpy(top, curr, curr_len);
                top[curr_len] = dna[c];
                top_len = curr_len + 1;
                top[top_len] = 0;
            }
            if (c == 0) { best_c = 0; best_prefix_prob = prefix_prob[0]; }
            else if (prefix_prob[c] > best_prefix_prob) { best_c = c; best_prefix_prob = prefix_prob[c]; }
        }
        if (best_prefix_prob < top_prob) break;
        curr[curr_len++] = dna[best_c];
        curr[curr_len] = 0;
        memcpy(alpha_prev, alphas + (size_t)best_c * T, sizeof(double) * T);
    }
    if (top_len + 1 > cap) { rc = PO_E_CAP; goto done; }
    memcpy(label, top, top_len + 1);
    *logp = top_prob;
    rc = top_len;
done:
    free(alpha_prev); free(alphas); free(ast); free(curr); free(top);
    return rc;
}

/* pair_prefix_search_log / _cy, prefix_search.py:247-385 (dense gamma, no envelope) */
/* Pair prefix search (prefix_search.py:247-385).  env == NULL: the dense gamma of the Python paths.  env != NULL
 * (U + 1 rows, inclusive column ranges): gamma from the envelope DP of Gamma.h:15-98, -inf outside the stored
 * ranges — the WORKING form of PairPrefixSearch.cpp:79-229, whose upstream version passes its gamma matrices by
 * value to pair_gamma_log_envelope_inplace (so they stay empty) and reads alpha_ast1[U] out of bounds; the prefix
 * probability is the sum over the cells (u, v), u < U, v < V, as in the Python paths. */
static int pair_prefix_search_impl(const double* y1, int U, const double* y2, int V, int C,
                                   int flavor, const int* env, char* label, int cap, double* logp) {
    static const char dna[] = "ACGT";
    const int A = C - 1;
    if (U < 1 || V < 1 || A < 1 || A > 4) return PO_E_ARG;
    const size_t W1 = (size_t)V + 1;
    const int M = U > V ? U : V;
    double* gm = (double*)malloc(sizeof(double) * (U + 1) * W1);
    double* a1p = (double*)malloc(sizeof(double) * U);
    double* a2p = (double*)malloc(sizeof(double) * V);
    double* a1s = (double*)malloc(sizeof(double) * U * A);
    double* a2s = (double*)malloc(sizeof(double) * V * A);
    double* ast1 = (double*)malloc(sizeof(double) * U);
    double* ast2 = (double*)malloc(sizeof(double) * V);
    double* flat = (double*)malloc(sizeof(double) * (size_t)U * V);
    char* curr = (char*)calloc(M + 4, 1);
    /* label_prob is a dict over every prefix ever scored; top_label = argmax over it, ties
     * resolved by insertion order (python max keeps the first maximum) */
    int rc = PO_E_NOMEM;
    size_t nlab = 0, caplab = 64;
    double* lab_p = (double*)malloc(sizeof(double) * caplab);
    char** lab_s = (char**)malloc(sizeof(char*) * caplab);
    if (!gm || !a1p || !a2p || !a1s || !a2s || !ast1 || !ast2 || !flat || !curr || !lab_p || !lab_s) goto done;
    if (env) { (void)pair_gamma_envelope_impl(y1, y2, env, U, V, C, gm); }
    else if (oracle_pair_gamma_dense(y1, U, y2, V, C, flavor, gm)) goto done;
    {
        double g = 0;
        for (int t = 0; t < U; ++t) g += y1[(size_t)t * C + A];
        double g2 = 0;
        for (int t = 0; t < V; ++t) g2 += y2[(size_t)t * C + A];
        lab_p[0] = g + g2;
        lab_s[0] = (char*)calloc(1, 1);
        nlab = 1;
    }
    size_t top = 0;
    int curr_len = 0, level = 0;
    oracle_forward_vec_log(-1, 0, y1, U, C, NULL, flavor, a1p);
    oracle_forward_vec_log(-1, 0, y2, V, C, NULL, flavor, a2p);
    const double g00 = gm[0];
    for (;;) {
        level++;
        int stop = 0;
        if (curr_len > M) stop = 1; /* 'Max search depth exceeded', :277-279 */
        double prefix_prob[4];
        for (int c = 0; c < A; ++c) {
            fw_no_gap(curr_len + 1, c, y1, U, C, a1p, ast1);
            fw_no_gap(curr_len + 1, c, y2, V, C, a2p, ast2);
            for (int u = 0; u < U; ++u)
                for (int v = 0; v < V; ++v) flat[(size_t)u * V + v] = ast1[u] + ast2[v] + gm[(size_t)(u + 1) * W1 + v + 1];
            prefix_prob[c] = seq_logsumexp_shift(flat, U * V, 1) - g00;
            oracle_forward_vec_log(c, level, y1, U, C, a1p, flavor, a1s + (size_t)c * U);
            oracle_forward_vec_log(c, level, y2, V, C, a2p, flavor, a2s + (size_t)c * V);
            if (nlab == caplab) {
                caplab *= 2;
                double* np_ = (double*)realloc(lab_p, sizeof(double) * caplab);
                char** ns_ = (char**)realloc(lab_s, sizeof(char*) * caplab);
                if (!np_ || !ns_) { if (np_) lab_p = np_; if (ns_) lab_s = ns_; goto done; }
                lab_p = np_; lab_s = ns_;
            }
            lab_p[nlab] = a1s[(size_t)c * U + U - 1] + a2s[(size_t)c * V + V - 1] - g00;
            lab_s[nlab] = (char*)malloc(curr_len + 2);
            if (!lab_s[nlab]) goto done;
            memcpy(lab_s[nlab], curr, curr_len);
            lab_s[nlab][curr_len] = dna[c];
            lab_s[nlab][curr_len + 1] = 0;
            nlab++;
        }
        int best_c = 0; /* max(prefix_prob.items(), key=value): first maximum wins */
        for (int c = 1; c < A; ++c) if (prefix_prob[c] > prefix_prob[best_c]) best_c = c;
        if (prefix_prob[best_c] < lab_p[top]) break;
        /* else-branch, :303-308: top_label = first maximum of label_prob in insertion order */
        { size_t bi = 0; for (size_t i = 1; i < nlab; ++i) if (lab_p[i] > lab_p[bi]) bi = i; top = bi; }
        curr[curr_len++] = dna[best_c];
        curr[curr_len] = 0;
        memcpy(a1p, a1s + (size_t)best_c * U, sizeof(double) * U);
        memcpy(a2p, a2s + (size_t)best_c * V, sizeof(double) * V);
        if (stop) break; /* stop_search was already set by the depth guard */
    }
    {
        int tl = (int)strlen(lab_s[top]);
        if (tl + 1 > cap) { rc = PO_E_CAP; goto done; }
        memcpy(label, lab_s[top], tl + 1);
        *logp = lab_p[top];
        rc = tl;
    }
done:
    for (size_t i = 0; i < nlab; ++i) free(lab_s[i]);
    free(lab_p); free(lab_s);
    free(gm); free(a1p); free(a2p); free(a1s); free(a2s); free(ast1); free(ast2); free(flat); free(curr);
    return rc;
}
int oracle_pair_prefix_search_log(const double* y1, int U, const double* y2, int V, int C,
                                  int flavor, char* label, int cap, double* logp) {
    return pair_prefix_search_impl(y1, U, y2, V, C, flavor, NULL, label, cap, logp);
}
int oracle_pair_prefix_search_log_env(const double* y1, int U, const double* y2, int V, int C, int flavor,
                                      const int* env, char* label, int cap, double* logp) {
    return pair_prefix_search_impl(y1, U, y2, V, C, flavor, env, label, cap, logp);
}

/* =====================================================================================
 * Needleman-Wunsch                                                  (align/align.pyx:29-178)
 * ===================================================================================== */
/* align.pyx:9-11 defaults; the functions take match / mismatch / gap_cost arguments upstream (align.pyx:29,100) */
static int g_nw_match = 2, g_nw_mismatch = -1, g_nw_gap = -1;
void oracle_set_nw_scores(int match, int mismatch, int gap_cost) { g_nw_match = match; g_nw_mismatch = mismatch; g_nw_gap = gap_cost; }
#define NW_MATCH g_nw_match
#define NW_MISMATCH g_nw_mismatch
#define NW_GAP g_nw_gap

static char py_index(const char* s, int len, int i, int* err) { /* python str[i] with wraparound */
    if (i < 0) i += len;
    if (i < 0 || i >= len) { *err = 1; return '?'; }
    return s[i];
}

/* shared traceback, align.pyx:56-95 == :137-174: every neighbour that equals the maximum is
 * taken in turn (no `break`), so one pass can emit up to three columns. */
typedef int (*dp_get_fn)(const void* ctx, int i, int j);
static int nw_traceback(const void* ctx, dp_get_fn get, const char* s1, int l1, const char* s2,
                        int l2, char* a1, char* a2, int cap) {
    int i = l1, j = l2, n = 0, err = 0;
#define EMIT(c1_, c2_) do { if (n + 1 >= cap) return PO_E_CAP; a1[n] = (c1_); a2[n] = (c2_); n++; } while (0)
    while (i > 0 && j > 0) {
        /* the trace-back calls scoring_function WITHOUT its match / mismatch arguments (align.pyx:66,143): the defaults
         * 2 / -1 whatever the fill used; gap_cost is the caller's */
        int sc = (py_index(s1, l1, i - 1, &err) == py_index(s2, l2, j - 1, &err)) ? 2 : -1;
        int cells[3] = {get(ctx, i - 1, j - 1) + sc, get(ctx, i - 1, j) + NW_GAP, get(ctx, i, j - 1) + NW_GAP};
        int mx = cells[0];
        if (cells[1] > mx) mx = cells[1];
        if (cells[2] > mx) mx = cells[2];
        for (int k = 0; k < 3; ++k) {
            if (cells[k] != mx) continue;
            if (k == 0) { i--; j--; EMIT(py_index(s1, l1, i, &err), py_index(s2, l2, j, &err)); }
            else if (k == 1) { i--; EMIT(py_index(s1, l1, i, &err), '-'); }
            else { j--; EMIT('-', py_index(s2, l2, j, &err)); }
        }
    }
    while (i > 0 || j > 0) {
        if (i > 0) { i--; EMIT(py_index(s1, l1, i, &err), '-'); }
        else if (j > 0) { j--; EMIT('-', py_index(s2, l2, j, &err)); }
    }
#undef EMIT
    if (err) return PO_E_ARG; /* IndexError in the reference */
    for (int k = 0; k < n / 2; ++k) { /* .reverse() */
        char t = a1[k]; a1[k] = a1[n - 1 - k]; a1[n - 1 - k] = t;
        t = a2[k]; a2[k] = a2[n - 1 - k]; a2[n - 1 - k] = t;
    }
    a1[n] = a2[n] = '\0';
    return n;
}

typedef struct { const int* dp; int w; } dense_ctx;
static int dense_get(const void* c, int i, int j) { const dense_ctx* d = (const dense_ctx*)c; return d->dp[(size_t)i * d->w + j]; }

int oracle_global_pair(const char* s1, int l1, const char* s2, int l2, char* a1, char* a2, int cap) {
    const int w = l2 + 1;
    int* dp = (int*)calloc((size_t)(l1 + 1) * w, sizeof(int));
    if (!dp) return PO_E_NOMEM;
    for (int i = 0; i <= l1; ++i) dp[(size_t)i * w] = NW_GAP * i;
    for (int j = 0; j <= l2; ++j) dp[j] = NW_GAP * j;
    for (int i = 1; i <= l1; ++i)
        for (int j = 1; j <= l2; ++j) {
            int a = dp[(size_t)(i - 1) * w + j - 1] + (s1[i - 1] == s2[j - 1] ? NW_MATCH : NW_MISMATCH);
            int b = dp[(size_t)(i - 1) * w + j] + NW_GAP;
            int c = dp[(size_t)i * w + j - 1] + NW_GAP;
            int m = a > b ? a : b;
            dp[(size_t)i * w + j] = m > c ? m : c;
        }
    dense_ctx ctx = {dp, w};
    int n = nw_traceback(&ctx, dense_get, s1, l1, s2, l2, a1, a2, cap);
    free(dp);
    return n;
}

typedef struct { int nrows; const int* start; const int* end; const size_t* off; const int* v; } band_ctx;
static int band_get(const void* c, int i, int j) { /* SparseMatrix<int>::get, default 0 */
    const band_ctx* b = (const band_ctx*)c;
    if (i < 0 || i >= b->nrows) return 0;
    if (j < b->start[i] || j > b->end[i]) return 0;
    return b->v[b->off[i] + (size_t)(j - b->start[i])];
}

int oracle_global_pair_banded(const char* s1, int l1, const char* s2, int l2, int band, char* a1,
                              char* a2, int cap) {
    /* align.pyx:100-178.  The boundary `set`s at :112-116 are no-ops (the matrix has no rows
     * yet); rows 0..l1-1 are pushed inside the fill loop; range(start, end) excludes `end`;
     * seq[i-1] / seq[j-1] wrap at index 0; out-of-band reads return 0. */
    int* start = (int*)malloc(sizeof(int) * (l1 ? l1 : 1));
    int* end = (int*)malloc(sizeof(int) * (l1 ? l1 : 1));
    size_t* off = (size_t*)malloc(sizeof(size_t) * (l1 + 1));
    int* v = NULL;
    int rc = PO_E_NOMEM, err = 0;
    if (!start || !end || !off) goto done;
    off[0] = 0;
    for (int i = 0; i < l1; ++i) {
        int center = (int)nearbyint((double)l2 / (double)l1 * (double)i); /* np.round: half-even */
        start[i] = center - band > 0 ? center - band : 0;
        end[i] = center + band < l2 - 1 ? center + band : l2 - 1;
        int w = end[i] - start[i] + 1;
        off[i + 1] = off[i] + (size_t)(w > 0 ? w : 0);
    }
    v = (int*)calloc(off[l1] ? off[l1] : 1, sizeof(int));
    if (!v) goto done;
    {
        band_ctx ctx = {0, start, end, off, v};
        for (int i = 0; i < l1; ++i) {
            ctx.nrows = i + 1; /* push_row before filling row i */
            for (int j = start[i]; j < end[i]; ++j) {
                int sc = (py_index(s1, l1, i - 1, &err) == py_index(s2, l2, j - 1, &err)) ? NW_MATCH : NW_MISMATCH;
                int a = band_get(&ctx, i - 1, j - 1) + sc;
                int b = band_get(&ctx, i - 1, j) + NW_GAP;
                int c = band_get(&ctx, i, j - 1) + NW_GAP;
                int m = a > b ? a : b;
                v[off[i] + (size_t)(j - start[i])] = m > c ? m : c;
            }
        }
        if (err) { rc = PO_E_ARG; goto done; }
        ctx.nrows = l1;
        rc = nw_traceback(&ctx, band_get, s1, l1, s2, l2, a1, a2, cap);
    }
done:
    free(start); free(end); free(off); free(v);
    return rc;
}

/* =====================================================================================
 * path -> signal mapping, alignment -> envelope     (pair_decode.py:114-142, envelope.py:5-87)
 * ===================================================================================== */
int oracle_sequence_mapping(const int* path, int T, int kind, int* out) {
    int n = 0;
    if (kind == PO_KIND_POREOVER) {
        for (int i = 0; i < T; ++i) if (path[i] < 4) out[n++] = i;
    } else if (kind == PO_KIND_FLIPFLOP) {
        for (int i = 0; i < T; ++i) if (i == 0 || path[i] != path[i - 1]) out[n++] = i;
    } else if (kind == PO_KIND_BONITO) {
        for (int i = 0; i < T; ++i) {
            int prev = path[i == 0 ? T - 1 : i - 1]; /* path[-1] wraps at i == 0 */
            if (path[i] == 4 || path[i] == prev) continue;
            out[n++] = i;
        }
    } else return PO_E_ARG;
    return n;
}

int oracle_build_envelope(int U, int V, const char* a1, const char* a2, int ncol, const int* s2s1,
                          int n1, const int* s2s2, int n2, int padding, int* env) {
    if (n1 < 1 || n2 < 1) return PO_E_ARG; /* sequence_to_signal[-1] IndexError upstream */
    for (int i = 0; i < 2 * U; ++i) env[i] = -1;
    int xi = -1, yi = -1;
    for (int k = 0; k < ncol; ++k) { /* get_alignment_columns, envelope.py:26-44 */
        if (a1[k] != '-') xi++;
        if (a2[k] != '-') yi++;
        int i1 = xi < 0 ? 0 : (xi > n1 - 1 ? n1 - 1 : xi);
        int i2 = yi < 0 ? 0 : (yi > n2 - 1 ? n2 - 1 : yi);
        int sx = s2s1[i1], ex = (i1 + 1 < n1) ? s2s1[i1 + 1] : U;
        int sy = s2s2[i2], ey = (i2 + 1 < n2) ? s2s2[i2 + 1] : V;
        for (int i = sx; i < ex; ++i) { /* add_block, envelope.py:5-17 */
            if (i < U) {
                if (sy < env[2 * i] || env[2 * i] < 0) env[2 * i] = sy;
                if (ey > env[2 * i + 1] || env[2 * i + 1] < 0) env[2 * i + 1] = ey;
            }
        }
    }
    for (int i = 0; i < U; ++i) { /* padding, :73-75 */
        env[2 * i] = (env[2 * i] - padding > 0) ? env[2 * i] - padding : 0;
        env[2 * i + 1] = (env[2 * i + 1] + padding < V) ? env[2 * i + 1] + padding : V;
    }
    int prev_end = 0; /* fix-ups, :78-85 (prev_end only moves inside the second `if`) */
    for (int i = 0; i < U; ++i) {
        if (env[2 * i] > env[2 * i + 1]) env[2 * i] = 0;
        if (env[2 * i] > prev_end) { env[2 * i] = prev_end; prev_end = env[2 * i + 1]; }
    }
    return U;
}

void oracle_diagonal_envelope(int U, int V, int width, int* env) { /* pair_decode.py:497-498 */
    for (int u = 0; u < U; ++u) {
        int c = (int)((double)u / (double)U * (double)V);
        env[2 * u] = c - width > 0 ? c - width : 0;
        env[2 * u + 1] = c + width < V ? c + width : V;
    }
}

/* =====================================================================================
 * pair_decode_helper stage chain, default flags                  (pair_decode.py:305-529)
 * ===================================================================================== */
int oracle_pair_decode(const double* y1, int U, const double* y2, int V, int C, int kind, int W,
                       int method, int padding, int full_alignment, char* seq1, char* seq2,
                       char* consensus, int cap, int* env_out, oracle_pair_summary* sm) {
    static const int model_of_kind[3] = {PO_MODEL_CTC, PO_MODEL_MERGE, PO_MODEL_FLIPFLOP};
    if (kind < 0 || kind > 2) return PO_E_ARG;
    int rc = PO_E_NOMEM;
    int* p1 = (int*)malloc(sizeof(int) * (U > 0 ? U : 1));
    int* p2 = (int*)malloc(sizeof(int) * (V > 0 ? V : 1));
    int* m1 = (int*)malloc(sizeof(int) * (U > 0 ? U : 1));
    int* m2 = (int*)malloc(sizeof(int) * (V > 0 ? V : 1));
    int* env = env_out ? env_out : (int*)malloc(sizeof(int) * 2 * (U > 0 ? U : 1));
    char *a1 = NULL, *a2 = NULL;
    if (!p1 || !p2 || !m1 || !m2 || !env) goto done;
    memset(sm, 0, sizeof(*sm));
    int l1 = oracle_viterbi_decode(y1, U, C, kind, NULL, p1, seq1, cap);
    int l2 = oracle_viterbi_decode(y2, V, C, kind, NULL, p2, seq2, cap);
    if (l1 < 0 || l2 < 0) { rc = l1 < 0 ? l1 : l2; goto done; }
    sm->len1 = l1; sm->len2 = l2;
    if (abs(l1 - l2) > 1000) { sm->skipped = 1; rc = PO_SKIP_LENGTH; goto done; }
    int n1 = oracle_sequence_mapping(p1, U, kind, m1);
    int n2 = oracle_sequence_mapping(p2, V, kind, m2);
    if (n1 != l1 || n2 != l2) { rc = PO_E_ARG; goto done; } /* assert, :379,382 */
    int acap = 3 * (l1 + l2) + 16;
    a1 = (char*)malloc(acap);
    a2 = (char*)malloc(acap);
    if (!a1 || !a2) goto done;
    int ncol = full_alignment ? oracle_global_pair(seq1, l1, seq2, l2, a1, a2, acap)
                              : oracle_global_pair_banded(seq1, l1, seq2, l2, 500, a1, a2, acap);
    if (ncol < 0) { rc = ncol; goto done; }
    int matches = 0;
    for (int k = 0; k < ncol; ++k) matches += (a1[k] == a2[k]);
    sm->ncol = ncol;
    sm->identity = (double)matches / (double)ncol;
    if (sm->identity < 0.5) { sm->skipped = 1; rc = PO_SKIP_IDENTITY; goto done; }
    rc = oracle_build_envelope(U, V, a1, a2, ncol, m1, n1, m2, n2, padding, env);
    if (rc < 0) goto done;
    rc = oracle_beam_search_2d(y1, U, y2, V, C, "ACGT", env, W, model_of_kind[kind], method, consensus, cap);
done:
    free(p1); free(p2); free(m1); free(m2); free(a1); free(a2);
    if (!env_out) free(env);
    return rc;
}
