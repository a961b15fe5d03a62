#!/bin/bash
# GPU box, round 6 (second half: the tree build and the ANN build changed, the search kernels did not): kernel traces of
# the final library -- the configs[4]-shape leg, the bench at the driver's arguments, one tree build at 1M / 10M as a
# timeline, one calcNormals at 1M / 10M as per-kernel totals -- summarised on the box into gpurun_out/keep6/ (copy to
# profiles/r06b_*).  The counter summaries of the search kernels (profiles/r06_*pmc*) stay: same kernels.
cd "$GRAFT_REPO_ROOT"; K=gpurun_out/keep6; rm -rf $K; mkdir -p $K
export TMPDIR=/tmp
( cd /tmp; timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/k6c5 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --no-cpu > $GRAFT_REPO_ROOT/$K/r06b_c5_bench_under_rocprof.json 2> /tmp/k6c5.err )
python tools/r6_kernel_totals.py /tmp/k6c5 > $K/r06b_c5_kernel_stats.csv
( cd /tmp; timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/k6b -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-small-scans --no-c5 > $GRAFT_REPO_ROOT/$K/r06b_bench_under_rocprof_s20_w5.json 2> /tmp/k6b.err )
python tools/r6_kernel_totals.py /tmp/k6b > $K/r06b_kernel_stats_s20_w5.csv
for n in 1000000 10000000; do timeout 200 bash tools/tree_trace.sh $n > $K/r06b_tree_build_timeline_$n.txt 2>&1; done
for n in 1000000 10000000; do timeout 300 bash tools/r5_normals_trace.sh $n > /dev/null 2>&1; cp gpurun_out/r5norm/trace_$n.txt $K/r06b_normals_kernel_totals_$n.txt; done
rm -rf gpurun_out/tt gpurun_out/r5norm
ls -la $K
