#!/bin/bash
# resource usage of ONE translation unit (default: the register-state pair kernel): tools/isa/ru_one.sh [file.hip] [extra hipcc flags]
f=${1:-poreover_amd/csrc/po_beam2d_reg.hip}; shift
cd $(dirname $0)/../..
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -S --cuda-device-only "$@" -o /tmp/ru_one.s $f 2>/dev/null
python3 - <<'PY'
import re,subprocess
txt=open('/tmp/ru_one.s').read()
meta=txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g=lambda k:(re.search(r"\.%s:\s+(\S+)"%k,blk) or [None,"0"])[1]
    dem=subprocess.run(["c++filt",g("name")],capture_output=True,text=True).stdout.strip()
    dem=re.sub(r"\(.*","",dem).replace("void ","").replace("(anonymous namespace)::","")
    print("%-45s vgpr %s sgpr %s sgpr_spill %s vgpr_spill %s scratch %s lds %s"%(dem,g("vgpr_count"),g("sgpr_count"),g("sgpr_spill_count"),g("vgpr_spill_count"),g("private_segment_fixed_size"),g("group_segment_fixed_size")))
PY
