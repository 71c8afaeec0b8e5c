#!/bin/bash
# several soaks side by side on one GPU (the soak is host-bound: the device is mostly idle), configurations from the command line
# usage: bash tools/soak_matrix.sh seconds "name copies VAR=value ..." ...
T=${1:-300}; shift
mkdir -p gpurun_out/soak; rm -f gpurun_out/soak/*.log
SEED=${SOAK_SEED0:-100}          # (SOAK_SEED0=1000: another set of random orders)
for spec in "$@"; do
  set -- $spec; name=$1; copies=$2; shift 2
  script=tools/soak_batch.py                 # SOAK_SCRIPT=tools/soak_inc.py in a spec: the incremental demo's soak instead of the batch path's
  for a in "$@"; do case $a in SOAK_SCRIPT=*) script=${a#SOAK_SCRIPT=};; esac; done
  for i in $(seq 1 $copies); do
    SEED=$((SEED + 1))
    env APRILSAM_AMD_PLAN_THREADS=2 "$@" timeout $((T + 120)) python $script $T $SEED > gpurun_out/soak/${name}_$i.log 2>&1 &
  done
done
wait
for spec in "$@"; do :; done
for f in gpurun_out/soak/*.log; do echo "== $f: $(tail -n 1 $f)"; grep "DIFFERENT\|POISON SEEN" $f | cut -c1-420; done
