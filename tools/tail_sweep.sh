#!/bin/bash
# chunk-taper sweep of the nwin=128 core; needs build/ab/tail.so = tools/mk.sh /root/repo/build/ab/tail.so -DHSS_TAIL_ENV
for cfg in "16 6" "16 0" "8 4" "32 6" "24 2" "40 10" "0 0"; do set -- $cfg; echo "tail4=$1 tail2=$2: $(HSSFSST_TAIL4=$1 HSSFSST_TAIL2=$2 AB_STEPS=100 AB_ROUNDS=3 python tools/ab_bench.py build/ab/tail.so | tail -1 | cut -c30-110)"; done
