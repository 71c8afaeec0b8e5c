#!/bin/bash
# several soaks side by side on one GPU (the soak is host-bound: the device is mostly idle), configurations from the command line
# usage: bash tools/soak_matrix.sh seconds "name copies VAR=value ..." ...
T=${1:-300}; shift
mkdir -p gpurun_out/soak; rm -f gpurun_out/soak/*.log
SEED=100
for spec in "$@"; do
  set -- $spec; name=$1; copies=$2; shift 2
  for i in $(seq 1 $copies); do
    SEED=$((SEED + 1))
    env APRILSAM_AMD_PLAN_THREADS=2 "$@" timeout $((T + 60)) python tools/soak_batch.py $T $SEED > gpurun_out/soak/${name}_$i.log 2>&1 &
  done
done
wait
for spec in "$@"; do :; done
for f in gpurun_out/soak/*.log; do echo "== $f: $(tail -n 1 $f)"; grep "DIFFERENT" $f | cut -c1-420; done
