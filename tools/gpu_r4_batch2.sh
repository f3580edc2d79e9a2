#!/bin/bash
# round 4, second GPU call: instruction rates at 1 - 8 waves per SIMD, A/B of the kept twiddle bases / early E_hd fetch / LDS table, new tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4b
tools/exp/valu_rates > gpurun_out/r4b/valu_rates.txt 2>&1; grep -E "v_fma_f32|v_pk_fma|v_fma_f64|v_cvt_f32|v_mov" gpurun_out/r4b/valu_rates.txt
bash tools/gpu_ab.sh r4b/ab base nr ehd nt
timeout 900 python -m pytest tests -m gpu -x -q -k "config3_first_steps or configs_4_and_5 or rx_golden or experiments or two_ranks or coupled or gpus_2" > gpurun_out/r4b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4b/pytest.log
tail -6 gpurun_out/r4b/pytest.log
