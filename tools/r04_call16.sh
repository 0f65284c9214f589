#!/bin/bash
# round 4, call 16: the whole GPU suite on the final tree, second run
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call16
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
grep -n "GRAPH CAPTURE\|graph mismatch probe\|hipGraph captures\|passed\|failed\|rc=" $out/tests.txt | cut -c1-2000
