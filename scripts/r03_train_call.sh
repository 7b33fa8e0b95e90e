#!/bin/bash
out=gpurun_out/r03_train; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > $out/suite.txt; grep -E "passed|failed|Error|assert" $out/suite.txt | head -8
python scripts/time_image_iter.py 2>&1 | tail -3
SLIDERS_TRAIN_NO_LN_FOLD=1 SLIDERS_TRAIN_UNFUSED_GEGLU=1 python scripts/time_image_iter.py 2>&1 | tail -2
python scripts/time_train_iter.py 2>&1 | tail -4
