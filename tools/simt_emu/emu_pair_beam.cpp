// TEST INFRASTRUCTURE ONLY — the row_col pair beam search of the register-state route (pre-pass, walk, beam2d_reg_kernel)
// executed on the CPU by the SIMT emulator, behind one C entry point for ctypes (tools/simt_emu/check_emu.py compares
// it with the oracle).  The kernels are the product's own sources, compiled with -DPO_EMU.
#include <hip/hip_runtime.h>

#include "../../poreover_amd/csrc/po_beam2d_pre.h"
#include "../../poreover_amd/csrc/po_beam2d_reg.hip"

extern "C" int emu_ring_pair_beam(const double* y1, const int64_t* y1_off, const double* y2, const int64_t* y2_off,
                                  const int32_t* env, int n, int C, int A, uint32_t alphabet, int W, char* seq,
                                  const int64_t* seq_off, int32_t* seq_len, int32_t* status, int blocks,
                                  unsigned long long* upd_count, int kernel, int model) {
    int64_t tr1 = y1_off[n] - y1_off[0], tr2 = y2_off[n] - y2_off[0], mr1 = 0, mr2 = 0;
    for (int i = 0; i < n; ++i) {
        mr1 = std::max<int64_t>(mr1, y1_off[i + 1] - y1_off[i]);
        mr2 = std::max<int64_t>(mr2, y2_off[i + 1] - y2_off[i]);
    }
    if (blocks <= 0) blocks = 1;
    X2Args a;
    memset(&a, 0, sizeof(a));
    a.y1 = y1; a.y1_off = y1_off; a.y2 = y2; a.y2_off = y2_off; a.env = env;
    a.n = n; a.A = A; a.W = W; a.C = C; a.alphabet = alphabet;
    a.seq = seq; a.seq_off = seq_off; a.seq_len = seq_len; a.status = status; a.use_pre_status = 0;
    std::vector<int> queue(64, 0), nmain(n, 0), envt(2 * (size_t)tr2 + 2, 0);
    std::vector<int2> meta(n);
    std::vector<int4> sched((size_t)tr2 + 1);
    std::vector<double> cum1((size_t)tr1 + 1), cum2((size_t)tr2 + 1);
    const int wide = (W > 6) ? 1 : 0;
    const size_t pool_bytes = po_reg_pool_bytes(model, wide);
    (void)kernel;
    // the library's slice pool, stood in for by host memory: one chunk, a slice per pair wave of the launch
    const int64_t WM = wide ? 12 : 6;
    const size_t arena_cap = (size_t)(1 + PO_A + (int64_t)PO_A * WM * 1024);
    const size_t slice_bytes = (pool_bytes + sizeof(int) * 3 * arena_cap + sizeof(int) * 2 * PO_A * (size_t)po_reg_ngl(wide) + 255) & ~size_t(255);
    std::vector<char> pool(slice_bytes * blocks, 0);
    std::vector<int> claim((size_t)blocks, 0);
    for (int i = 0; i < blocks; ++i) claim[(size_t)i] = i;
    unsigned tickets[2] = {0u, 0u};
    a.queue = queue.data(); a.meta = meta.data(); a.nmain = nmain.data(); a.sched = sched.data(); a.envt = envt.data();
    a.cum1 = cum1.data(); a.cum2 = cum2.data();
    a.pool = nullptr; a.pool_bytes = pool_bytes; a.arena = nullptr; a.arena_cap = (long long)arena_cap;
    for (int c = 0; c < 8; ++c) a.slice_chunk[c] = nullptr;
    a.slice_chunk[0] = pool.data(); a.slice_spc_log2 = 30; a.nslices = blocks; a.slice_bytes = slice_bytes; a.slice_claim = claim.data();
    a.slice_tickets = tickets; a.persist = 1; a.defer_count = nullptr; a.starve = 0;
    a.dbg = nullptr; a.upd_count = upd_count; a.defer_odd = 0; a.need_mono = 1; a.order = nullptr;
    a.wgstate = nullptr; a.magic = 0;
    a.pre_vcols = (int)std::min<int64_t>(mr2, 6144);
    a.ngl = po_reg_ngl(wide);
    a.no_cum = 1;
    a.chain_scan = getenv("EMU_CHAIN_SCAN") ? atoi(getenv("EMU_CHAIN_SCAN")) : 0;
    for (int i = 0; i < n; ++i) status[i] = PO_OK;
    const bool vb = getenv("EMU_VERBOSE") != nullptr;
#ifdef PO_EMU_SHADOW
    po_emu_shadow_reset();
#endif
    if (vb) fprintf(stderr, "[emu] prepass\n");
    if (model == PO_MODEL_CTC) hipLaunchKernelGGL(beam2d_prepass_kernel<PO_MODEL_CTC>, dim3(n), dim3(64), 0, nullptr, a);
    else if (model == PO_MODEL_MERGE) hipLaunchKernelGGL(beam2d_prepass_kernel<PO_MODEL_MERGE>, dim3(n), dim3(64), 0, nullptr, a);
    else hipLaunchKernelGGL(beam2d_prepass_kernel<PO_MODEL_FLIPFLOP>, dim3(n), dim3(64), 0, nullptr, a);
    if (vb) fprintf(stderr, "[emu] walk\n");
    hipLaunchKernelGGL(beam2d_walk_kernel, dim3(n), dim3(64), 0, nullptr, a);
    if (vb) fprintf(stderr, "[emu] main kernel\n");
    po_reg_launch(&a, blocks, model, wide, nullptr);
    int deferred = 0;
    for (int i = 0; i < n; ++i)
        if (meta[i].y == X2_DEFERRED) { deferred++; status[i] = -100; }   // (the product hands these to beam2d_kernel)
    return deferred;
}
