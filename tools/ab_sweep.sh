#!/bin/bash
# tools/ab_sweep.sh "lib1 lib2 ..." "walk values" logn bits -- run ON THE GPU BOX: tools/size_sweep.py cells under several builds
# of the library (HEXL_AMD_LIB) and settings of the "walk14" tuning key, alternating, twice.
LIBS=$1; WALKS=$2; LOGN=$3; BITS=$4
for rep in 1 2; do for l in $LIBS; do for w in $WALKS; do
  echo "## $l walk14=$w"
  HEXL_AMD_LIB=$PWD/$l WALK=$w SWEEP_LOGN_MIN=$LOGN SWEEP_LOGN_MAX=$LOGN SWEEP_BITS=$BITS python - <<PY | grep "^| 2"
import os, sys, runpy
sys.path.insert(0, os.getcwd())
import hexl_amd as hx
hx.set_tuning("walk14", int(os.environ["WALK"]))
runpy.run_path("tools/size_sweep.py", run_name="__main__")
PY
done; done; done
