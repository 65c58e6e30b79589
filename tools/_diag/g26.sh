#!/bin/bash
python tools/spark_curve.py 60 f16 || exit 1
CMU_SPARK_CELLS=0 CMU_SPARK_C1_TILES=0 python tools/spark_curve.py 60 f16 || exit 1
CMU_SPARK_TILES=0 python tools/spark_curve.py 60 f16 || exit 1
