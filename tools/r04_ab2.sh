#!/bin/bash
# round 4, second GPU call: parity subset on the partition-liveness binning + new exchange, shard step, PMC of the compare kernel (r03 vs tree)
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py tests/test_gpu_configs.py tests/test_dist_gpu.py tests/test_jni_binding.py -m gpu -x -q > gpurun_out/r04/pytest_gpu_2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu_2.log
tail -4 gpurun_out/r04/pytest_gpu_2.log
for lib in flashfry_amd/lib/ab/r03.so flashfry_amd/lib/libflashfry_hip.so; do
  for n in 1 2 4 8; do
    echo "== $lib shards $n" | tee -a gpurun_out/r04/shard_step.txt
    FFH_LIBRARY=$PWD/$lib timeout 300 python tools/shard_step.py --shards $n --rank $((n / 2)) --comm 2>&1 | grep '^{' | tee -a gpurun_out/r04/shard_step.txt
  done
done
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
for lib in flashfry_amd/lib/ab/r03.so flashfry_amd/lib/libflashfry_hip.so; do
  echo "== PMC $lib" | tee -a gpurun_out/r04/pmc_compare.txt
  FFH_LIBRARY=$GRAFT_REPO_ROOT/$lib KREGEX="k_compare<" bash tools/pmc_prepare.sh 2>&1 | tee -a gpurun_out/r04/pmc_compare.txt
done
