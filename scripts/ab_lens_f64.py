"""Scratch (GPU box): the fp64 curved-lens table of bench.py's extra.lens_fermat_f64 (1024 elements x 512^2 targets of configs[3]),
one library per process (RTUS_LIB); HIP events around 5 launches; checksum and max |dT| against variants/librtus_prev.so's table
when a reference file is given."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import rtus, bench
L = rtus.lib()
W = bench.lens_inputs(0, 1)
mk = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
nf = 512 * 512
txe, tze, txf, tzf = mk(W["xe"]), mk(W["ze"]), mk(W["xf"][:nf]), mk(W["zf"][:nf])
out = torch.empty((1024, nf), dtype=torch.float64, device="cuda")
lens = rtus.Params().lens()
run = lambda: L.rtus_tt_lens_dev(C.byref(lens), -rtus.ALPHA_MAX, rtus.ALPHA_MAX, txe.data_ptr(), tze.data_ptr(), 1024, txf.data_ptr(), tzf.data_ptr(), nf,
                                 out.data_ptr(), None, None)
for _ in range(2): assert run() == 0
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 5)
tag = os.environ.get("RTUS_LIB", "tree").split("/")[-1]
print(tag, "%.4f ms" % best, "checksum %.17g" % float(out.sum()), flush=True)
sub = out[::37, ::101].cpu().numpy()
path = "/tmp/lens64_ref.npy"
if tag != "tree":
    np.save(path, sub)
elif os.path.exists(path):
    print("max |dT| against the other library's table (subsample): %.3e s" % np.nanmax(np.abs(sub - np.load(path))))
