#!/bin/bash
# GPU box, round 4: lazy scan moves -- parity, the 1-GPU round with and without, the 8-rank rehearsal, SQ counters of the link launch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "lazy or lum_links or company or config4 or rank_count or clpairs or gapx or lum6DQuat or two_ranks or slam_glue or match_graph or deferred or moves" > gpurun_out/r4a/pytest.log 2>&1; tail -5 gpurun_out/r4a/pytest.log
for lz in 1 0; do
  TDTK_LAZY_MOVES=$lz timeout 600 python bench.py --workload graphslam --steps 10 --warmup 3 > gpurun_out/r4a/gs_lazy$lz.json 2> gpurun_out/r4a/gs_lazy$lz.err
  python -c "import json;d=json.load(open('gpurun_out/r4a/gs_lazy$lz.json'));print('lazy=$lz ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'])"
done
timeout 900 python tools/gs_shard_probe.py > gpurun_out/r4a/shard.log 2>&1; tail -7 gpurun_out/r4a/shard.log
cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --workload graphslam --steps 4 --warmup 2"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4a
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq1 -o p -- $CMD > $OUT/sq1.json 2> $OUT/sq1.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq2 -o p -- $CMD > $OUT/sq2.json 2> $OUT/sq2.err
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.json 2> $OUT/fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.json 2> $OUT/write.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, glob, json
out = {}
for p in ("sq1", "sq2", "fetch", "write"):
    for fn in glob.glob("gpurun_out/r4a/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            if "k_search_refill_multi" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = {"avg": sum(v) / len(v), "n": len(v)}
json.dump(out, open("gpurun_out/r4a/gs_counters.json", "w"), indent=1)
print(json.dumps(out))
PY
find gpurun_out/r4a -name "*.csv" -size +1M -delete
