#!/bin/bash
# GPU box: the ICP probe at the driver's arguments and the defaults, twice each (kernel selection from the environment)
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
for rep in 1 2; do N=1000000 K=20 W=5 run A=1; N=1000000 K=100 W=10 run A=1; done
