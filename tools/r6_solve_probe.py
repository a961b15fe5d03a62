"""GPU box: duration of k_final_solve with its full arithmetic and with an early exit (a NaN among the sums), 200 calls each --
run under rocprofv3 --kernel-trace; tools/r6_solve_probe.sh prints the two groups."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t = importlib.import_module("3dtk_amd")
capi = importlib.import_module("3dtk_amd._capi")
L = t.lib()
rng = np.random.default_rng(1)
d = rng.uniform(-1000, 1000, (20000, 3))
T = t.EulerToMatrix4([1.5, -2.0, 0.7], [0.002, -0.003, 0.005])
R = np.array([[T[0], T[4], T[8]], [T[1], T[5], T[9]], [T[2], T[6], T[10]]])
m = d @ R.T + T[12:15] + rng.normal(0, 1, d.shape)
acc = np.zeros(17); acc[0] = len(d); acc[1] = ((m - d) ** 2).sum(); acc[2:5] = m.sum(0); acc[5:8] = d.sum(0)
acc[8:17] = (m[:, :, None] * d[:, None, :]).sum(0).reshape(9)
bad = acc.copy(); bad[9] = np.nan
xf = np.empty(16); rms = C.c_double(0); st = C.c_int(0); sh = np.zeros(3)
for a in (acc, bad):
    for _ in range(200):
        capi.check(L.tdtk_icp_device_solve(0, capi.dptr(a), capi.dptr(sh), capi.dptr(xf), C.byref(rms), C.byref(st)))
    print("status", st.value)
