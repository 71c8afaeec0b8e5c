#!/bin/bash
# Collect the rocprofv3 evidence of one round on the MI355X box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01
# Outputs under gpurun_out/prof_<tag>/ ; tools/pmc_aggregate.py turns them into the files kept in profiles/.
# Counter passes are separate runs (one --pmc set each, no trace domains besides the kernel trace), without hipGraph
# replay (counter collection over graph replays does not terminate on ROCm 7.2) and each under its own timeout.
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 200 --warmup 10 --no-lattice --no-inc --no-cpu-baseline"
SHORT="python $ROOT/bench.py --steps 10 --warmup 2 --no-lattice --no-inc --no-cpu-baseline"
LAT="python $ROOT/tools/lattice_big.py 316 2"
rocprofv3 -L > $OUT/counters_available.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  APRILSAM_AMD_USE_GRAPH=0 timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- $SHORT > $OUT/pmc_$C.log 2>&1
done
# MFMA utilisation of the wide-supernode path: 100k lattice (config 4), counters per dispatch
APRILSAM_AMD_USE_GRAPH=0 timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- $LAT > $OUT/pmc_mfma.log 2>&1
APRILSAM_AMD_USE_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lattice -- $LAT > $OUT/stats_lattice.log 2>&1
# config 5 on one GPU (1M poses): kernel stats + MFMA counters of the wide-supernode path
APRILSAM_AMD_USE_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lattice1m -- python $ROOT/tools/lattice_big.py 1000 1 > $OUT/stats_lattice1m.log 2>&1
APRILSAM_AMD_USE_GRAPH=0 timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma1m -- python $ROOT/tools/lattice_big.py 1000 1 > $OUT/pmc_mfma1m.log 2>&1
# HBM traffic of the bandwidth-bound kernels on the lattices (k_linearize, k_assemble_big, k_backsolve_*): FETCH_SIZE / WRITE_SIZE,
# one pass per counter and workload
for C in FETCH_SIZE WRITE_SIZE; do
  APRILSAM_AMD_USE_GRAPH=0 timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_lat316 -- $LAT > $OUT/pmc_${C}_lat316.log 2>&1
  APRILSAM_AMD_USE_GRAPH=0 timeout 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_lat1000 -- python $ROOT/tools/lattice_big.py 1000 1 > $OUT/pmc_${C}_lat1000.log 2>&1
done
# phase stamps of the outer-block chain kernel (root front of the 100k lattice)
timeout 120 python $ROOT/tools/chain_times.py 316 > $OUT/chain_times_lattice100k.txt 2>&1
# config 3: the incremental demo (first 1500 poses)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_inc -- python $ROOT/tools/inc_demo.py 3500 > $OUT/stats_inc.log 2>&1
python $ROOT/tools/trace_medians.py $OUT/stats_inc > $OUT/inc_trace_medians.txt 2>&1
# ... its host-side split and the phases inside k_inc_one (wall-clock stamps), where its time goes, the first call of a process, the planner
APRILSAM_AMD_INC_PROFILE=2 timeout 100 python $ROOT/tools/inc_demo.py 3500 > $OUT/inc_profile.txt 2>&1
timeout 100 python $ROOT/tools/inc_slowest.py > $OUT/inc_slowest.txt 2>&1
# ... per class of step (fronts regenerated / updated), with and without the low-rank updates; the demo's --batch_update_only mode
APRILSAM_AMD_INC_PROFILE=1 timeout 100 python $ROOT/tools/inc_steps.py 3500 > $OUT/inc_steps.txt 2>&1
APRILSAM_AMD_INC_UPDATE=0 timeout 100 python $ROOT/tools/inc_steps.py 3500 > $OUT/inc_steps_no_update.txt 2>&1
timeout 100 python $ROOT/tools/batch_only.py 1200 > $OUT/batch_only.txt 2>&1
timeout 100 python $ROOT/tools/first_call.py --count-first > $OUT/first_call.txt 2>&1
APRILSAM_AMD_PLAN_PROFILE=1 timeout 100 python $ROOT/tools/plan_time.py > $OUT/plan_time.txt 2>&1
timeout 60 $ROOT/tools/ubench/launch_lat > $OUT/ubench_launch_lat.txt 2>&1
# instruction-rate micro-benchmarks the roofline discussion quotes
timeout 60 $ROOT/tools/ubench/mfma_f64 > $OUT/ubench_mfma_f64.txt 2>&1
timeout 60 $ROOT/tools/ubench/valu_lat > $OUT/ubench_valu_lat.txt 2>&1
cd $ROOT
python tools/pmc_aggregate.py $OUT $TAG > $OUT/aggregate.log 2>&1
tail -5 $OUT/aggregate.log
