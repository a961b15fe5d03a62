#!/bin/bash
# GPU box: kernel-selection sweep of the ICP loop (one process per configuration)
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} 10 2>&1 | tail -1; }
python -c "
import importlib, ctypes as C
t = importlib.import_module('3dtk_amd'); L = t.lib()
g = C.c_double()
for kind, b in ((0, 1 << 30), (0, 1 << 31), (1, 16 << 20), (1, 8 << 20)):
    L.tdtk_measure_bandwidth(0, kind, b, 5, C.byref(g)); print('bandwidth kind', kind, 'bytes', b, '->', round(g.value, 1), 'GB/s')
"
for N in ${SIZES:-1000000 4000000}; do
  export N
  run TDTK_FUSE_SUMS=0
  run TDTK_FUSE_SUMS=1
  for q in 128 192 384 512; do run TDTK_FUSE_SUMS=0 TDTK_REFILL_QPW=$q; done
  for q in 128 512; do run TDTK_FUSE_SUMS=1 TDTK_REFILL_QPW=$q; done
  for th in 8 32; do run TDTK_FUSE_SUMS=0 TDTK_REFILL_THRESH=$th; run TDTK_FUSE_SUMS=1 TDTK_REFILL_THRESH=$th; done
  run TDTK_FUSE_SUMS=0 TDTK_SEARCH_VARIANT=4
  run TDTK_FUSE_SUMS=0 TDTK_SEARCH_VARIANT=8
done
