#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace + stats, then HBM counters in separate --pmc passes
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no --pmc together with traces).
cd /tmp && export TMPDIR=/tmp
: "${GRAFT_REPO_ROOT:?tools/profile.sh runs on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT  # a stale pass must not be summarized with this one
mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.csrc_fingerprint())" > $OUT/csrc_sha16.txt
for W in "$@"; do
  S=100; [ "$W" != "funnel_1e4" ] && S=10
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python3 $R/bench.py --steps $S --warmup 5 --min-seconds 0 --no-cpu-baseline --no-extra --workload $W > $OUT/bench_trace_$W.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$W -- python3 $R/bench.py --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-extra --workload $W > $OUT/bench_fetch_$W.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$W -- python3 $R/bench.py --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-extra --workload $W > $OUT/bench_write_$W.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq_$W -- python3 $R/bench.py --steps 10 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-extra --workload $W > $OUT/bench_sq_$W.log 2>&1
done
python3 $R/bench.py --steps 300 --warmup 30 > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -1 $OUT/bench_full.json | cut -c1-200
