// Host-side device facts, PER DEVICE and safe to ask from several host threads at once.  The in-process multi-device
// pipeline (po_multi_pair_decode) plans launch geometry on one host thread per device: CU count, memory budget and
// kernel occupancy were process-wide statics filled from whichever device a thread happened to be on first — a data
// race on first use, and wrong on a node with unlike (or partitioned) devices.
#pragma once
#ifndef PO_EMU
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

constexpr int PO_MAX_DEVICES = 64;

inline int po_cur_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PO_MAX_DEVICES) dev = 0;
    return dev;
}

struct PoDevInfo { int cus; size_t mem; };
inline const PoDevInfo& po_dev_info() {
    static PoDevInfo info[PO_MAX_DEVICES];
    static std::once_flag once[PO_MAX_DEVICES];
    const int dev = po_cur_device();
    std::call_once(once[dev], [dev] {
        hipDeviceProp_t p;
        PoDevInfo x{0, 0};
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) { x.cus = p.multiProcessorCount; x.mem = p.totalGlobalMem; }
        if (x.cus <= 0) x.cus = 256;
        if (!x.mem) x.mem = (size_t)288 << 30;
        info[dev] = x;
    });
    return info[dev];
}

// One small integer per (device, key), computed on first use by `compute` (an occupancy query): the value is a pure
// function of device and kernel, so two threads racing to fill a slot store the same number.
template <int NKEYS>
struct PoPerDeviceCache {
    std::atomic<int> v[PO_MAX_DEVICES][NKEYS];
    PoPerDeviceCache() { for (auto& d : v) for (auto& x : d) x.store(0, std::memory_order_relaxed); }
    template <class F>
    int get(int key, F compute) {
        std::atomic<int>& slot = v[po_cur_device()][key];
        int x = slot.load(std::memory_order_relaxed);
        if (!x) { x = compute(); slot.store(x, std::memory_order_relaxed); }
        return x;
    }
};
#endif
