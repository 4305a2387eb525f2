#!/bin/bash
# rocprofv3 evidence for the loop a user calls: the native muse! loop (muse_run) at configs[1] -- kernel trace + stats of
# tools/runloop_bench.py, then HBM and SQ counters in separate --pmc passes (as tools/profile.sh), then the stamp breakdown
# of the iteration kernel (tools/stamps_run.py, needs the -DMUSE_STAMPS build).  Output: gpurun_out/prof_run/.
cd /tmp && export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?tools/profile_run.sh runs on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_run
rm -rf $OUT
mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.csrc_fingerprint())" > $OUT/csrc_sha16.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/runloop_bench.py host > $OUT/runloop_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/runloop_bench.py host once > $OUT/runloop_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/runloop_bench.py host once > $OUT/runloop_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/runloop_bench.py host once > $OUT/runloop_sq.log 2>&1
python3 $R/tools/runloop_bench.py > $OUT/runloop.log 2>&1
[ -f $R/museinference.jl_amd/libmuse_hip_stamps.so ] && python3 $R/tools/stamps_run.py > $OUT/stamps_run.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -r head -5
cat $OUT/runloop.log $OUT/stamps_run.log 2>/dev/null | grep -v amdgpu.ids
