#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call12
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_deepseek.py tests/test_gpu_ep.py -x -q > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
tail -15 $out/tests.txt
timeout 600 python -m pytest tests/test_gpu_production_shapes.py -x -q -k "v2_lite" >> $out/tests.txt 2>&1
tail -3 $out/tests.txt
echo "== v2lite: three launches for the experts (CHITU_MOE_FUSE_SILU=0)" > $out/extras.txt
CHITU_MOE_FUSE_SILU=0 timeout 300 python tools/run_extra.py v2lite 32 2>&1 | tail -1 >> $out/extras.txt
echo "== v2lite now (KV row in the absorb launch; GEMM1 with SiLU + quant epilogue at bs 16)" >> $out/extras.txt
timeout 300 python tools/run_extra.py v2lite 32 2>&1 | tail -1 >> $out/extras.txt
echo "== ep8 rank now / three launches" >> $out/extras.txt
timeout 300 python tools/run_extra.py ep8 32 2>&1 | tail -1 >> $out/extras.txt
CHITU_MOE_FUSE_SILU=0 timeout 300 python tools/run_extra.py ep8 32 2>&1 | tail -1 >> $out/extras.txt
cat $out/extras.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_v2
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_v2 -o t -- python $GRAFT_REPO_ROOT/tools/run_extra.py v2lite 8 > $out/run_v2.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/prof_v2/t_results.db --last-fraction 0.3 > $out/v2lite_kerneltrace.txt
