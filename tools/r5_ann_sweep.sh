for at in 16 32 48 64; do echo "refill_at=$at"; TDTK_ANN_REFILL_AT=$at python tools/normals_probe.py --n 1000000 --reps 4 2>&1 | tail -2; done
echo "one kernel"; TDTK_LAB=1 TDTK_ANN_SPLIT=0 python tools/normals_probe.py --n 1000000 --reps 4 2>&1 | tail -2
