echo "two kernels (product)"; python tools/normals_probe.py --n 1000000 --reps 4 2>&1 | tail -2; echo "two kernels (lab)"; TDTK_LIB=lab python tools/normals_probe.py --n 1000000 --reps 4 2>&1 | tail -2
echo "one kernel"; TDTK_LIB=lab TDTK_ANN_SPLIT=0 python tools/normals_probe.py --n 1000000 --reps 4 2>&1 | tail -2
