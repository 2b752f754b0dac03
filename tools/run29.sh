#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r3s; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
run() { local label=$1; shift
  env INSAR_TAPE=0 "$@" timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" "$label" | tee -a "$OUT/marginal.txt"
}
for round in 1 2; do
  run base X=1
  run no_wgrad_gemms INSAR_EXPERIMENT_SKIP=insar_wgrad_conv3,insar_wgrad
  run no_wgrad_at_all INSAR_EXPERIMENT_SKIP=insar_wgrad_conv3,insar_wgrad,insar_wgrad_reduce,insar_wgrad_fold,insar_conv3x3_small_wgrad
  run no_bwd_apply INSAR_EXPERIMENT_SKIP=insar_bnrelu_bwd_apply,insar_bnrelu_bwd_apply_pool,insar_bnrelu_bwd_apply_outc
  run no_coef INSAR_EXPERIMENT_SKIP=insar_bnse_bwd_coef,insar_bn_bwd_coef
  run no_bwd_reduce INSAR_EXPERIMENT_SKIP=insar_bnrelu_bwd_reduce,insar_bnrelu_bwd_reduce_pool,insar_bnrelu_bwd_reduce_outc
  run no_fwd_apply INSAR_EXPERIMENT_SKIP=insar_bn_relu_apply,insar_bn_relu_apply_pool_arg,insar_bn_relu_apply_outc
  run no_finalize INSAR_EXPERIMENT_SKIP=insar_bn_finalize
  run no_squeeze_excite INSAR_EXPERIMENT_SKIP=insar_se_squeeze,insar_se_excite
  run no_adam INSAR_EXPERIMENT_SKIP=insar_adam_step
  run no_dgrad_gemms_deep INSAR_EXPERIMENT_SKIP=insar_igemm
  run no_flat_c64 INSAR_EXPERIMENT_SKIP=insar_conv3x3_flat,insar_conv3x3_flat_bstat,insar_conv3x3_c64,insar_conv3x3_c64_bstat
done
echo done
