mkdir -p gpurun_out/r04
python -m pytest tests -x -q -m gpu > gpurun_out/r04/t.log 2>&1; grep -E "passed|failed|Error|error|assert" gpurun_out/r04/t.log | tail -12
python tools/probe_first_forward.py 2>&1 | grep -E "forward" | head -4
python tools/probe_first_forward.py ant-int-pot-flint 2>&1 | grep -E "forward" | head -4
ANTQ_BATCH_CALIB=0 python tools/probe_first_forward.py ant-int-pot-flint 2>&1 | grep -E "forward" | head -4
