#!/bin/bash
# usage (GPU box): tests/gpu_ab_solver.sh LIB_A LIB_B [reps] — back end alone at L = 5000 and 20000 under two builds of the
# library, alternating on the same box
A=$1; B=$2; reps=${3:-3}
for r in $(seq $reps); do
  for lib in $A $B; do
    for L in 5000 20000; do
      QTR_LIB=$lib timeout 120 python $GRAFT_REPO_ROOT/tests/gpu_solver_prof.py $L 30 2>/dev/null | tail -1 | sed "s#^#$(basename $lib) #" | cut -c1-260
    done
  done
done
