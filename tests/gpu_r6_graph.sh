#!/bin/bash
# usage (GPU box): tests/gpu_r6_graph.sh — the three forms of the consistency-graph kernel under the test build's QTR_GRAPH knob:
# bit matrix against the oracle at awkward sizes and noise bounds (tests/gpu_graph_bench.py check), then the graph stage's time
R=$GRAFT_REPO_ROOT; cd $R
T=$R/quatro_amd/libquatro_hip_testengines.so
echo "== mfma: check"; QTR_LIB=$T QTR_GRAPH=mfma timeout 600 python tests/gpu_graph_bench.py check 2>&1 | grep -E "MISMATCH|!=|graph check|Error|error" | head -20
for g in mfma strips tiles mfma strips; do echo "== $g: time"; QTR_LIB=$T QTR_GRAPH=$g timeout 300 python tests/gpu_graph_bench.py time 2>&1 | grep "graph stage"; done
echo "== mfma: the L = 20000 solver test, the back-end tests"; QTR_LIB=$T QTR_GRAPH=mfma timeout 600 python -m pytest tests -x -q -m gpu -k "L20000 or solve or clique or core or second_run" 2>&1 | tail -2
