#!/bin/bash
# round 4, eighth GPU call: device-side coupling over a one-rank communicator, test durations of the whole suite
cd "$(dirname "$0")/.."
O=gpurun_out/r4i; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "device_side_coupling or coupled or rccl_communicator" > $O/pytest_coupling.log 2>&1; echo "pytest rc=$?" >> $O/pytest_coupling.log; tail -15 $O/pytest_coupling.log
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep -E "passed|failed|s call|s setup" $O/pytest.log | head -40
