// TEST INFRASTRUCTURE ONLY: a host-side stand-in for <hip/hip_runtime.h> that lets a wave-synchronous HIP kernel of
// this repo be compiled with g++ and executed on the CPU, one fibre per lane (tools/simt_emu/README.md).  It exists
// because the development container has no GPU: kernel LOGIC (tables, cross-lane exchanges, LDS hand-overs) is
// checked against the oracle here before a GPU-minute is spent.  Nothing under poreover_amd/ uses it; the product
// library is built by hipcc from the same sources without PO_EMU.
//
// Model: a workgroup is a set of fibres scheduled round-robin on one OS thread.  A cross-lane operation (__shfl,
// __ballot, readlane, wave barrier) is a rendezvous of the 64 fibres of a wave and must be reached by all of them from
// the same source line (divergent use aborts with both line numbers); __syncthreads is a rendezvous of the block.
// A fibre runs uninterrupted between rendezvous points, so plain loads / stores / "atomics" need no locking.
#pragma once
#define PO_EMU 1
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __noinline__ __attribute__((noinline))

struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint3_ { unsigned x, y, z; };
struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
#define hipSuccess 0
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 0; return 1; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n); return 0; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
#define hipMemcpyDeviceToHost 0
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }

namespace emu {
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    uint3_ tid{0, 0, 0};
};
struct Rv {   // rendezvous of a group of fibres
    int arrived = 0, gen = 0, line = -1;
};
extern Fiber* g_cur;
extern uint3_ g_bid, g_bdim, g_gdim;
extern int g_nlive_block;
extern unsigned long long g_xch[1024][2];
void yield();
void rendezvous_wave(int line);
void dump_wave_lines();
void rendezvous_block(int line);
int wave_live(int wave);
unsigned long long wave_group();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
long long clock_ticks();
}  // namespace emu

#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_bid)
#define blockDim (emu::g_bdim)
#define gridDim (emu::g_gdim)

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) emu::launch((grid), (block), [=]() { kern(__VA_ARGS__); })

// ---- cross-lane operations (wave = 64 consecutive threads of the block)
namespace emu {
template <class T>
static inline T xch_get(int tid_in_block) { T v; memcpy(&v, &g_xch[tid_in_block][0], sizeof(T)); return v; }
template <class T>
static inline T shfl(T x, int src, int line) {
    static_assert(sizeof(T) <= 16, "shfl payload");
    const int me = (int)g_cur->tid.x, base = me & ~63;
    memcpy(&g_xch[me][0], &x, sizeof(T));
    rendezvous_wave(line);
    if (!((wave_group() >> (src & 63)) & 1ull)) {
        fprintf(stderr, "[simt_emu] line %d: thread %d reads lane %d, which is not executing this operation\n", line, me, src & 63);
        dump_wave_lines();
        abort();
    }
    const T r = xch_get<T>(base + (src & 63));
    rendezvous_wave(line);
    return r;
}
static inline unsigned long long ballot(bool p, int line) {
    const int me = (int)g_cur->tid.x, base = me & ~63;
    g_xch[me][0] = p ? 1ull : 0ull;
    rendezvous_wave(line);
    const unsigned long long grp = wave_group();   // (lanes outside the branch, or gone, do not vote)
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (((grp >> l) & 1ull) && g_xch[base + l][0]) m |= 1ull << l;
    rendezvous_wave(line);
    return m;
}
}  // namespace emu
// DPP (v_mov_b32_dpp through __builtin_amdgcn_update_dpp): the data-parallel-primitive lane patterns of GFX9 — quad_perm,
// row_shl / row_shr / row_ror, wave_shr:1 & co, row_mirror, row_half_mirror, row_bcast15 / 31 — with row_mask, bank_mask and
// bound_ctrl.  A source lane outside the row (or not executing) is invalid: the result is 0 with bound_ctrl, `old` without.
namespace emu {
static inline int dpp_src(int lane, int ctrl) {   // source lane, or -1: invalid
    const int row = lane & ~15, l = lane & 15;
    if (ctrl >= 0 && ctrl <= 0xFF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; return (l + n < 16) ? lane + n : -1; }
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; return (l - n >= 0) ? lane - n : -1; }
    if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; return row | ((l - n) & 15); }
    if (ctrl == 0x130) return (lane + 1 < 64) ? lane + 1 : -1;
    if (ctrl == 0x134) return (lane + 1) & 63;
    if (ctrl == 0x138) return (lane - 1 >= 0) ? lane - 1 : -1;
    if (ctrl == 0x13C) return (lane - 1) & 63;
    if (ctrl == 0x140) return row | (15 - l);
    if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));
    if (ctrl == 0x142) return (lane >= 16) ? (row - 1) : -1;
    if (ctrl == 0x143) return (lane >= 32) ? 31 : -1;
    fprintf(stderr, "[simt_emu] unknown dpp_ctrl 0x%x\n", ctrl);
    abort();
}
static inline int dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int line) {
    const int me = (int)g_cur->tid.x, base = me & ~63, lane = me & 63;
    memcpy(&g_xch[me][0], &src, sizeof(int));
    rendezvous_wave(line);
    int r = old;
    if (((row_mask >> (lane >> 4)) & 1) && ((bank_mask >> ((lane & 15) >> 2)) & 1)) {
        const int sl = dpp_src(lane, ctrl);
        const bool ok = sl >= 0 && ((wave_group() >> sl) & 1ull);
        r = ok ? xch_get<int>(base + sl) : (bound_ctrl ? 0 : old);
    }
    rendezvous_wave(line);
    return r;
}
}  // namespace emu
#define __builtin_amdgcn_update_dpp(old_, src_, ctrl_, rm_, bm_, bc_) emu::dpp((int)(old_), (int)(src_), (ctrl_), (rm_), (bm_), (bc_), __LINE__)
#define __shfl(v_, l_) emu::shfl((v_), (l_), __LINE__)
static inline int emu_lane_() { return (int)(emu::g_cur->tid.x & 63); }
#define __shfl_xor(v_, m_) emu::shfl((v_), (int)(emu_lane_() ^ (m_)), __LINE__)
#define __ballot(p_) emu::ballot((p_), __LINE__)
// v_permlane32_swap_b32 (gfx950): the upper half of the first operand and the lower half of the second change places; the builtin
// returns both registers
struct emu_u2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
static inline emu_u2 emu_permlane32_swap(unsigned a, unsigned b, int line) {
    const int lane = emu_lane_();
    const unsigned a_other = emu::shfl(a, lane ^ 32, line), b_other = emu::shfl(b, lane ^ 32, line);
    emu_u2 r;
    r.v[0] = (lane < 32) ? a : b_other;
    r.v[1] = (lane < 32) ? a_other : b;
    return r;
}
#define __builtin_amdgcn_permlane32_swap(a_, b_, fi_, bc_) emu_permlane32_swap((unsigned)(a_), (unsigned)(b_), __LINE__)
#define __builtin_amdgcn_readlane(v_, l_) emu::shfl((int)(v_), (l_), __LINE__)
#define __builtin_amdgcn_readfirstlane(v_) emu::shfl((int)(v_), 0, __LINE__)
#define __builtin_amdgcn_ds_bpermute(a_, v_) emu::shfl((int)(v_), ((a_) >> 2), __LINE__)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_sched_barrier(m_) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu::rendezvous_wave(__LINE__)
#define __builtin_amdgcn_s_sleep(n) emu::yield()
#define __builtin_amdgcn_s_barrier() emu::rendezvous_block(__LINE__)
#define __syncthreads() emu::rendezvous_block(__LINE__)
static inline int emu_syncthreads_or(int p, int line) {
    static int acc[2];
    static int phase = 0;
    const int ph = phase;
    if (p) acc[ph] = 1;
    emu::rendezvous_block(line);
    const int r = acc[ph];
    emu::rendezvous_block(line);
    if (emu::g_cur->tid.x == 0) { acc[ph] = 0; phase ^= 1; }
    emu::rendezvous_block(line);
    return r;
}
#define __syncthreads_or(p_) emu_syncthreads_or((p_), __LINE__)

// ---- scalar helpers
static inline int __double2hiint(double x) { unsigned long long u; memcpy(&u, &x, 8); return (int)(unsigned)(u >> 32); }
static inline int __double2loint(double x) { unsigned long long u; memcpy(&u, &x, 8); return (int)(unsigned)u; }
static inline double __hiloint2double(int hi, int lo) {
    const unsigned long long u = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    double x; memcpy(&x, &u, 8); return x;
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline long long wall_clock64() { return emu::clock_ticks(); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
static inline size_t max(size_t a, size_t b) { return a > b ? a : b; }

// ---- "atomics": a fibre is never preempted between rendezvous points
template <class T> static inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { const T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T* p, T v) { const T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { const T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T c, T v) { const T o = *p; if (o == c) *p = v; return o; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// -DPO_EMU_SHADOW: a tag per store address beside the emulated store (what rounds 1 - 4 kept IN the store), for checking the
// presence bookkeeping of the tag-free entries against it
#ifdef PO_EMU_SHADOW
#include <unordered_map>
static inline std::unordered_map<unsigned long long, unsigned long long>& po_emu_shadow_map() {
    static std::unordered_map<unsigned long long, unsigned long long> m;
    return m;
}
static inline void po_emu_shadow_reset() { po_emu_shadow_map().clear(); }   // (per emulated launch: tags of an earlier launch's pairs must not hit)
static inline void po_emu_shadow_put(unsigned long long key, unsigned long long tag) { po_emu_shadow_map()[key] = tag; }
static inline unsigned long long po_emu_shadow_get(unsigned long long key) {
    auto& m = po_emu_shadow_map();
    auto it = m.find(key);
    return it == m.end() ? ~0ull : it->second;
}
#endif

