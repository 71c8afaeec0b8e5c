#!/bin/bash
# every full-size stress sweep side by side on one GPU (they are host-bound: the reference runs beside this library):
#   bash tools/stress_all.sh > gpurun_out/stress_all.txt       (one gpurun call; bounded slices of the same sweeps run in tests/test_gpu_sweeps.py)
mkdir -p gpurun_out/stress; rm -f gpurun_out/stress/*.log
run() { name=$1; shift; ( timeout 2000 python "$@" > gpurun_out/stress/$name.log 2>&1; echo "[rc $?]" >> gpurun_out/stress/$name.log ) & }
run growing_batch tools/stress_growing_batch.py
run incremental tools/stress_incremental.py
run lattices tools/stress_lattices.py
run long_incremental tools/stress_long_incremental.py
run random_graphs tools/stress_random_graphs.py
run random_graphs_guard tools/stress_random_graphs.py 500 512
run sharded tools/stress_sharded.py
run structured_graphs tools/stress_structured_graphs.py
wait
for f in gpurun_out/stress/*.log; do
  n=$(wc -l < $f); echo "== $(basename $f .log): $((n - 1)) lines, last: $(tail -n 2 $f | head -n 1 | cut -c1-160)  $(tail -n 1 $f)"
done
for f in long_incremental incremental; do echo; echo "--- $f, every case:"; grep -v "^\[rc" gpurun_out/stress/$f.log | cut -c1-200; done
