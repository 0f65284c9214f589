#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call10
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_moe_align.py tests/test_gpu_mixtral.py tests/test_gpu_ep.py -x -q > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
tail -5 $out/tests.txt
timeout 600 python -m pytest tests/test_gpu_production_shapes.py tests/test_gpu_deepseek.py -x -q -k "v2_lite or v2lite" >> $out/tests.txt 2>&1
tail -3 $out/tests.txt
echo "== v2lite before (three-launch experts, ticket route+align)" > $out/extras.txt
CHITU_MOE_FUSE_SILU=0 CHITU_DEBUG_OPTIONS=gate_ticket=1 timeout 300 python tools/run_extra.py v2lite 32 2>&1 | tail -1 >> $out/extras.txt
echo "== v2lite now" >> $out/extras.txt
timeout 300 python tools/run_extra.py v2lite 32 2>&1 | tail -1 >> $out/extras.txt
echo "== mixtral before (ticket route+align)" >> $out/extras.txt
CHITU_DEBUG_OPTIONS=gate_ticket=1 timeout 300 python tools/run_extra.py mixtral 32 2>&1 | tail -1 >> $out/extras.txt
echo "== mixtral now" >> $out/extras.txt
timeout 300 python tools/run_extra.py mixtral 32 2>&1 | tail -1 >> $out/extras.txt
cat $out/extras.txt
