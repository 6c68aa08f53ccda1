#!/bin/bash
# Copies one collection of profiles/collect_r6.sh (gpurun_out/TAG) into the judged files profiles/r6_* and rebuilds the derived
# summaries:   bash profiles/publish_r6.sh TAG      (the timing build must have been present for the two stamp files)
TAG=${1:?tag}; O=gpurun_out/$TAG; P=profiles
set -e
python -c "import json,sys; json.loads(open('$O/bench.json').read().strip().splitlines()[-1])"   # (the bench line parses)
python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
open('$P/r6_bench.json', 'w').write(json.dumps(d) + '\n')  # (the line as printed)
PY
for f in kernel_stats timeline batch_kernel_stats dense_step_kernel_stats dense_solver_kernel_stats solver5k_kernel_stats conn20k_kernel_stats \
         l5k_kernel_stats kernel_stamps recheck_stamps gpu_tests; do [ -f $O/$f.txt ] && cp $O/$f.txt $P/r6_$f.txt; done
for w in seq batch dense; do cp $O/pmc_$w.json $P/r6_pmc_$w.json; done
python $P/make_r6_fpfh.py $P/r6_pmc_seq_before.json $P/r6_pmc_seq.json $P/r6_pmc_dense_before.json $P/r6_pmc_dense.json > $P/r6_pmc_fpfh.json
python $P/make_r6_nn.py $O
ls -la $P/r6_* | cut -c24-
