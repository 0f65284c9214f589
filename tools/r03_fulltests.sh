#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03_fulltests}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
tail -40 $out/tests.txt
