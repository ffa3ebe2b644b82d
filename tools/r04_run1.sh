cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04a/bench200.json 2>gpurun_out/r04a/bench200.err
python tools/microbench/sgemm_yardstick.py > gpurun_out/r04a/sgemm.txt 2>&1
./tools/microbench/mfma_mix 20 > gpurun_out/r04a/mfma_mix_20.txt 2>&1
./tools/microbench/mfma_mix 300 > gpurun_out/r04a/mfma_mix_300.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04a/kt -- python $GRAFT_REPO_ROOT/tools/microbench/sgemm_yardstick.py > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r04a/kt -name "*kernel_stats*" | head; 
for f in $(find $GRAFT_REPO_ROOT/gpurun_out/r04a/kt -name "*kernel_stats.csv"); do head -12 $f | cut -c1-300; done
cat $GRAFT_REPO_ROOT/gpurun_out/r04a/sgemm.txt
tail -c 600 $GRAFT_REPO_ROOT/gpurun_out/r04a/bench200.json
