#!/bin/bash
# round 5, batch 7: stage kernels against the general kernel at chip-filling sizes, GPU fuzz and soak on the final sources
cd "$(dirname "$0")/.."
REPO=$PWD; O=$REPO/gpurun_out/r5_b7; mkdir -p $O
SSF_LIB=$REPO/opticommpy_amd/libssf_hip_exp.so timeout 1200 python tests/tools/split_check.py > $O/split_check.txt 2>&1; echo "split_check rc=$?" | tee -a $O/split_check.txt
grep -v WARNING $O/split_check.txt | tail -14
timeout 900 python tests/tools/fuzz_gpu.py 300 511 near > $O/fuzz_a.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz_a.txt; grep -v WARNING $O/fuzz_a.txt | tail -3
timeout 600 python tests/tools/soak_gpu.py > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/soak.txt; tail -4 $O/soak.txt
timeout 900 python -m pytest tests/test_experiments.py tests/test_coupled_gpu.py tests/test_gpu_tx.py -m gpu -q --timeout 800 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
