python -m pytest tests/test_model_gpu.py -q -x -k "grouped" 2>&1 | tail -3
python tools/config_sweep.py 2>/dev/null | grep "cfg"
