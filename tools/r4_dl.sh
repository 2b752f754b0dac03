set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4g; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 $R/tools/deeplab_unit_times.py 2>&1 | tee $OUT/deeplab_units.txt
