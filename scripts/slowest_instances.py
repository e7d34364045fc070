import os, sys
import numpy as np
sys.path.insert(0, "pantr-cdc2023-experiments_amd")
import pantr_amd as pa
d = np.load("tests/golden/mpc_hanging_chain_N60_warm300.npz")
x0 = np.vstack([d["x0"], d["x0"]]); u0 = np.vstack([d["guess"], np.zeros_like(d["guess"])])
b = pa.PantrBatch("hanging_chain", 60, 600, opts=pa.DODO_PANTR_OPTS)
for _ in range(2):
    u, st = b.solve(x0, u0)
print("kernel ms", b.last_kernel_ms())
o = np.argsort(-st["elapsed_s"])[:8]
for i in o:
    print(i, "cold" if i >= 300 else "warm", "elapsed %.1f ms"%(st["elapsed_s"][i]*1e3), "iters", st["iterations"][i], "hv", st["n_hess_psi_prod"][i], "us/hv %.1f"%(st["elapsed_s"][i]/st["n_hess_psi_prod"][i]*1e6))
print("sum hv", st["n_hess_psi_prod"].sum(), "sum iters", st["iterations"].sum(), "mean elapsed %.2f"%(st["elapsed_s"].mean()*1e3))
