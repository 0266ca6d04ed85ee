timeout -k 10 1100 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$" > gpurun_out/t1.log; grep -n "FAILED\|passed\|failed" gpurun_out/t1.log | head -40
