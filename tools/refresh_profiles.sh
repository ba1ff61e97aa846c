#!/bin/bash
# Round-end measurement refresh on the GPU box: the default bench line, its rocprofv3 summaries, the microbenchmarks
# (forward / backward / warp) and their kernel trace.  Everything lands in gpurun_out/profile_<tag>/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/profile_$TAG; mkdir -p $O; cd $R
# the PMC passes first, so that the bench line below quotes THIS round's memory-side traffic (bench.py reads the newest
# profiles/*_traffic.json; the copy into profiles/ here is in the box's scratch checkout -- commit it from gpurun_out/)
bash tools/profile_bench.sh $TAG > $O/profile_bench.log 2>&1
cp $O/${TAG}_traffic.json $R/profiles/ 2>/dev/null
python bench.py 2> $O/bench_stderr.log | tail -1 > $O/${TAG}_bench.json
cat $O/${TAG}_bench.json
(python tools/microbench.py --iters 30; python tools/microbench.py --iters 10 --config multiviewx; python tools/microbench.py --iters 5 --config stress16) 2>&1 | grep -v amdgpu.ids > $O/${TAG}_microbench.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/mb_trace -o t -- python $R/tools/microbench.py --iters 10 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/mb_fetch -o p -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/mb_write -o p -- python $R/tools/microbench.py --iters 3 > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $O/mb_trace/t_results.db --filter mvdetr > $O/${TAG}_microbench_kernel_stats.txt
# HBM-side bytes of every kernel of the microbenchmark (forward / backward / warp), separate --pmc passes;
# FETCH_SIZE is KB and reports half the bytes of 16-byte-per-lane reads on gfx950 (see ${TAG}_traffic.json)
python tools/rocpd_summary.py $O/mb_fetch/p_results.db --filter mvdetr > $O/${TAG}_microbench_pmc_fetch.txt
python tools/rocpd_summary.py $O/mb_write/p_results.db --filter mvdetr > $O/${TAG}_microbench_pmc_write.txt
rm -rf $O/mb_trace $O/mb_fetch $O/mb_write
cat $O/${TAG}_microbench.txt; cat $O/${TAG}_traffic.json
