#!/bin/bash
# kernel resources (VGPRs, SGPRs, scratch, LDS) of one csrc/*.hip, read from the code object's metadata: scripts/kres.sh rtus_lens_fermat [extra flags]
f=$1; shift
cd "$(dirname "$0")/../ray-tracing-ultrasound_amd/csrc"
extra=""
case $f in rtus_shoot|rtus_solve) extra="-ffp-contract=off -mllvm -disable-machine-licm";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-fast-math -fno-slp-vectorize $extra "$@" --cuda-device-only -S $f.hip -o /tmp/$f.s
python3 - /tmp/$f.s <<'PY'
import sys,re
name=None;rows={}
for ln in open(sys.argv[1]):
    ln=ln.strip()
    if ln.startswith(".name:"): name=ln.split()[1]
    for key in (".vgpr_count:",".sgpr_count:",".vgpr_spill_count:",".sgpr_spill_count:",".private_segment_fixed_size:",".group_segment_fixed_size:"):
        if ln.startswith(key) and name: rows.setdefault(name,{})[key[1:-1]]=int(ln.split()[1])
import subprocess
for k,v in rows.items():
    d=subprocess.run(["c++filt",k],capture_output=True,text=True).stdout.strip()
    print(d[:110], v)
PY
