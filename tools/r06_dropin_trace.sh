#!/bin/bash
# the soft-fp8 drop-in run under a per-op watchdog, on a fresh box, BEFORE anything else has
# touched torch's GEMM libraries in this call.  Needs gpurun_stage/reference (an uncommitted copy, removed afterwards).
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_dropin; mkdir -p $out
export CHITU_REFERENCE_DIR=$PWD/gpurun_stage/reference PYTHONDONTWRITEBYTECODE=1
{
echo "# fresh box, first process of the call: first-call costs of torch's bf16 GEMM path (hipBLASLt / rocBLAS), seconds"
timeout -k 5 400 python - <<'PY'
import time
t0 = time.time()
import torch, torch.nn.functional as F
print(f"import torch {time.time()-t0:.2f}", flush=True)
t = time.time(); x = torch.zeros(8, device="cuda"); torch.cuda.synchronize(); print(f"cuda init {time.time()-t:.2f}", flush=True)
for (m, n, k) in [(7, 768, 512), (7, 512, 2048), (2, 768, 512), (7, 1024, 512), (7, 16, 512)]:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16); w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    t = time.time(); y = F.linear(a, w); torch.cuda.synchronize(); print(f"F.linear bf16 M={m} N={n} K={k} first call {time.time()-t:.3f}", flush=True)
    t = time.time(); y = F.linear(a, w); torch.cuda.synchronize(); print(f"   second call {time.time()-t:.5f}", flush=True)
PY
echo "rc=$?"
} > $out/first_call_costs.txt 2>&1
for soft in 1 0; do
  start=$(date +%s)
  DROPIN_TRACE=1 REF_MASTER_PORT=2957$soft timeout -k 5 300 python tests/dropin_worker.py $CHITU_REFERENCE_DIR $soft > $out/trace_soft$soft.txt 2>&1
  echo "rc=$? wall=$(( $(date +%s) - start )) s" >> $out/trace_soft$soft.txt
done
for run in 1 2; do
  timeout -k 5 500 python -m pytest tests/test_gpu_reference_dropin.py -x -q -s > $out/pytest_run$run.txt 2>&1
  echo "rc=$?" >> $out/pytest_run$run.txt
done
grep -c "falling back" $out/trace_soft1.txt
tail -3 $out/first_call_costs.txt $out/trace_soft1.txt $out/trace_soft0.txt $out/pytest_run1.txt $out/pytest_run2.txt
