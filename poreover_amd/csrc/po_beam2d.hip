// placeholder until the pair kernels land (replaced below in this round)
#include "po_device.h"
extern "C" size_t po_beam2d_ws_bytes_impl(int, int64_t, int64_t, int64_t, int64_t, int, int, int, int) { return 256; }
extern "C" int po_launch_beam2d(const double*, const int64_t*, const double*, const int64_t*, const int32_t*, int, int,
                     int, uint32_t, int, int, int, char*, const int64_t*, int32_t*, int32_t*, void*, size_t, hipStream_t) { return PO_E_UNSUPPORTED; }
extern "C" size_t po_pair_ws_bytes_impl(int, int64_t, int64_t, int64_t, int64_t, int, const po_pair_options*) { return 256; }
extern "C" int po_launch_pair_decode(const double*, const int64_t*, const double*, const int64_t*, int, int,
                          const po_pair_options*, char*, const int64_t*, int32_t*, int32_t*, double*, int32_t*,
                          char*, const int64_t*, int32_t*, int32_t*, void*, size_t, hipStream_t) { return PO_E_UNSUPPORTED; }
