import os, sys, subprocess, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
G = '/root/repo/tests/golden'
z = np.load(os.path.join(G, "dat_scans.npz"))
def rf(p, rmax):
    d2 = (p * p).sum(1); return np.ascontiguousarray(p[d2 < rmax * rmax])
pts = [rf(z["scan%03d" % k], 500.0) for k in range(3)]
fin, fout = '/tmp/in.bin', '/tmp/out.bin'
with open(fin, 'wb') as f:
    f.write(np.int32(3).tobytes())
    for k in range(3):
        f.write(np.int32(len(pts[k])).tobytes()); f.write(np.ascontiguousarray(z["pose%03d" % k], dtype=np.float64).tobytes()); f.write(pts[k].tobytes())
exe = '/root/repo/adapters/harness/_bin/slam_glue_harness'
for rnd in ("5", "1"):
    for rep in range(3):
        r = subprocess.run([exe, "doicp", fin, fout, "1", "-1", "25.0", "50", "1e-5", rnd, "42"], capture_output=True, text=True, env=dict(os.environ, TDTK_HARNESS_DEBUG="1"))
        print(rnd, rep, r.stdout.strip().replace("\n", " | "), r.stderr[-200:])
