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
# both native loops: "host" = muse_run (one map_score_kernel launch per iteration), "dev" = muse_run_device (ONE muse_loop_kernel
# launch per 30-iteration call -- what muse() runs)
for L in host dev; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$L -- python3 $R/tools/runloop_bench.py $L > $OUT/runloop_trace_$L.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$L -- python3 $R/tools/runloop_bench.py $L once > $OUT/runloop_fetch_$L.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$L -- python3 $R/tools/runloop_bench.py $L once > $OUT/runloop_write_$L.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq_$L -- python3 $R/tools/runloop_bench.py $L once > $OUT/runloop_sq_$L.log 2>&1
done
python3 $R/tools/runloop_bench.py > $OUT/runloop.log 2>&1
STAMPS=$R/museinference.jl_amd/libmuse_hip_stamps.so   # (python museinference.jl_amd/build.py --stamps, before the gpurun call)
for f in $R/museinference.jl_amd/csrc/* $R/include/muse_hip.h; do
  if [ -f $STAMPS ] && [ $f -nt $STAMPS ]; then echo "libmuse_hip_stamps.so is older than $f: rebuild it (build.py --stamps); no stamp breakdown"; STAMPS=/nonexistent; fi
done
if [ -f $STAMPS ]; then
  python3 $R/tools/stamps_run.py > $OUT/stamps_run_host.log 2>&1
  python3 $R/tools/stamps_run.py 10000 1 512 4 dev > $OUT/stamps_run_dev.log 2>&1
fi
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -r head -5
cat $OUT/runloop.log $OUT/stamps_run_host.log $OUT/stamps_run_dev.log 2>/dev/null | grep -v amdgpu.ids
