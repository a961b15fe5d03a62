R=$GRAFT_REPO_ROOT
bash tools/profile_bench.sh r2F > /dev/null 2>&1
bash tools/profile_graphslam.sh r2F > /dev/null 2>&1
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2F/bench_steps20.json 2> gpurun_out/r2F/bench_steps20.err
python bench.py > gpurun_out/r2F/bench.json 2> gpurun_out/r2F/bench.err
python tools/summarize_profiles.py r2F r02 > gpurun_out/r2F/summ.log 2>&1
python tools/summarize_graphslam_profile.py r2F r02 > gpurun_out/r2F/summ_gs.log 2>&1
mkdir -p gpurun_out/r2F/profiles; cp profiles/r02_* gpurun_out/r2F/profiles/
for d in stats fetch write sq1 sq2 sq3 tcc gs gs_fetch gs_write; do rm -rf gpurun_out/r2F/$d; done
du -sh gpurun_out/r2F; tail -5 gpurun_out/r2F/summ.log; tail -3 gpurun_out/r2F/summ_gs.log
