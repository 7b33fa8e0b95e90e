#!/bin/bash
out=gpurun_out/r03_small; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > $out/suite.txt; cat $out/suite.txt | head -2
python scripts/time_train_iter.py --breakdown > $out/pieces.txt 2>&1; grep -A14 "denoise pass:" $out/pieces.txt | head -16; head -6 $out/pieces.txt
python scripts/time_image_iter.py 2>&1 | tail -3
