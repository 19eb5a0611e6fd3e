#!/bin/bash
# Developer tool (GPU box): per-op table of the trunk for several A/B builds.  usage: bash tools/ab.sh libA.so libB.so ...
for lib in "$@"; do
  HIPAC_LIB_NAME=$lib python tools/opbench.py bf16 20 2>&1 | tail -2 | sed "s/^lib=[^ ]* //" | cut -c1-700; echo " <- $lib"
done
