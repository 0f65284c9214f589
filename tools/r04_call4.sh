#!/bin/bash
# round 4, call 4: the whole GPU suite with the staged-cure probe armed (tests/conftest.py)
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call4
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
grep -n "GRAPH CAPTURE\|graph mismatch probe\|hipGraph captures\|passed\|failed\|rc=" $out/tests.txt | cut -c1-6000
