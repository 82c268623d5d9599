#!/usr/bin/env python3
"""Where an epoch of fqi_epochs_kernel goes: shader-clock stamps of the first 4 epochs of the last launch.
   GRLX_FQI_STAMPS=1 python3 tools/fqi_stamps.py <replicas> <batch_size> <epochs>"""
import ctypes as C, os, sys
os.environ["GRLX_FQI_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd
R, n, ep = (int(v) for v in sys.argv[1:4])
cfg = grl_amd.pendulum_fqi_config(R, batch_size=n, iterations=2, epochs=ep, max_batches=2)
r = grl_amd.FqiRunner(cfg, np.arange(1, R + 1))
r.run_batch(); r.run_batch(); r.sync()
out = np.zeros(R * 16 * 4 * 8, np.uint64)
lib = grl_amd.capi.load()
lib.grlx_fqi_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
assert lib.grlx_fqi_debug_stamps(r._ctx, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size) == 0
s = out.reshape(R, 16, 4, 8).astype(np.int64)
names = ["chunk loop", "publish + meet", "trees + step", "block barrier"]
for e in range(1, 4):
    d = np.diff(s[:, :, e, :5], axis=-1)
    print(f"epoch {e}: " + ", ".join(f"{nm} mean {d[..., k].mean():.0f} max {d[..., k].max()} min {d[..., k].min()}" for k, nm in enumerate(names)),
          f"| epoch period {np.mean(s[:, :, e, 0] - s[:, :, e - 1, 0]):.0f} clocks")
    f = s[:, :, e, :]
    print(f"   first chunk: prologue + loads issued {np.mean(f[..., 5] - f[..., 0]):.0f}, forward pass {np.mean(f[..., 6] - f[..., 5]):.0f}, "
          f"backward + tile writes {np.mean(f[..., 7] - f[..., 6]):.0f}")
