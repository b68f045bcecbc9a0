#!/bin/bash
# TFM change gate: its tests, time per launch, then L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate passes)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/tfm; rm -rf $OUT; mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_tfm.py -x -q -m gpu 2>&1 | tail -15 || exit 1
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from importlib import import_module
dev_api = import_module("ray-tracing-ultrasound_amd.device")
t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
n_el, n_t, fs = 64, 2048, 50e6
xe = (np.arange(n_el) - (n_el - 1) / 2.0) * 0.6e-3
xs, zs = np.meshgrid(np.linspace(-0.02, 0.02, 256), np.linspace(0.025, 0.045, 256))
tt = dev_api.tt_layers_dev([0.020], [2330.0, 1483.0], t64(xe), t64(np.zeros(n_el)), t64(xs.ravel()), t64(zs.ravel()))
fmc = torch.randn((n_el, n_el, n_t), dtype=torch.float32, device="cuda")
img = torch.empty(xs.size, dtype=torch.float32, device="cuda")
for _ in range(3): dev_api.tfm_dev(fmc, fs, tt, out=img)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): dev_api.tfm_dev(fmc, fs, tt, out=img)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print(f"tfm 64x64x2048 -> 256x256: {best*1e3:.1f} us per launch")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/rd -o rd --output-format csv -- python3 $ROOT/scripts/run_consumers_once.py > $OUT/rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/wr -o wr --output-format csv -- python3 $ROOT/scripts/run_consumers_once.py > $OUT/wr.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("rd", "wr"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "rtus_" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "mean per dispatch (KiB)", round(sum(v) / len(v), 1))
PY
