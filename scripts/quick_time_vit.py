"""Device-resident timing of the Viterbi kernel (HIP events via po_profile)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poreover_amd import _lib
from poreover_amd.batch import pack_rows
from poreover_amd.synth import synth_pair
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = [synth_pair(i, T=4000)[0] for i in range(64)]
y, off, Cc = pack_rows([base[i % 64] for i in range(n)])
dev = torch.device("cuda")
dy, doff = torch.from_numpy(y).to(dev), torch.from_numpy(off).to(dev)
rows = int(off[-1])
dseq = torch.empty(rows, dtype=torch.uint8, device=dev); dlen = torch.zeros(n, dtype=torch.int32, device=dev)
dst = torch.zeros(n, dtype=torch.int32, device=dev); dmap = torch.empty(rows, dtype=torch.int32, device=dev)
dpath = torch.empty(rows, dtype=torch.int8, device=dev)
ws = torch.empty(256, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for with_path in (False, True):
    for rep in range(3):
        lib.po_profile_enable(1); lib.po_profile_reset()
        for _ in range(5):
            _lib.check(lib.po_viterbi_batch(dy.data_ptr(), doff.data_ptr(), n, Cc, b"ACGT", 0, dpath.data_ptr() if with_path else None,
                                            dseq.data_ptr(), doff.data_ptr(), dlen.data_ptr(), dmap.data_ptr(), dst.data_ptr(),
                                            ws.data_ptr(), 256, stream), "vit")
        torch.cuda.synchronize()
        ms, cnt = C.c_double(), C.c_int64(); lib.po_profile_get(_lib.K_VITERBI, C.byref(ms), C.byref(cnt))
    nb = int(dlen.sum().item())
    byts = 8.0 * Cc * rows + 5.0 * nb + (rows if with_path else 0)
    avg = ms.value / cnt.value
    print("viterbi n=%d path=%s: %.3f ms/launch  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)  %.1f Mreads/s" % (
        n, with_path, avg, byts / (avg * 1e-3) / 1e9, byts / (avg * 1e-3) / 1e9 / 80.0, n / (avg * 1e-3) / 1e6))
