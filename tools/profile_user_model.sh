#!/bin/bash
# rocprofv3 evidence for the user-supplied example model (models/cubic.h at theta = 0, N = 10^4, 512 sims per launch): kernel
# trace + stats, then the HBM counters in separate --pmc passes, laid out like tools/profile.sh's output so that
# tools/summarize_profiles.py <tag> condenses it (workload "cubic_1e4").
cd /tmp && export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?tools/profile_user_model.sh runs on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT
mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.csrc_fingerprint())" > $OUT/csrc_sha16.txt
export THETA=0.0
W=cubic_1e4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python3 $R/tools/user_model_bench.py 40 "cubic atol 1e-6" > $OUT/bench_trace_$W.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$W -- python3 $R/tools/user_model_bench.py 8 "cubic atol 1e-6" > $OUT/bench_fetch_$W.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$W -- python3 $R/tools/user_model_bench.py 8 "cubic atol 1e-6" > $OUT/bench_write_$W.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq_$W -- python3 $R/tools/user_model_bench.py 8 "cubic atol 1e-6" > $OUT/bench_sq_$W.log 2>&1
tail -1 $OUT/bench_trace_$W.log
