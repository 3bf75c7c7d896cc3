#!/bin/bash
# GPU box: serial per-kernel timeline of the recording stage for experiment builds (make variant) of the refinement kernel
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/rgbdslam_v2_amd/librgbdfe_$v.so
  [ "$v" = "base" ] && lib=$GRAFT_REPO_ROOT/rgbdslam_v2_amd/librgbdfe.so
  for noise in 0.01 0.002; do
    echo "== $v noise $noise: $(RGBDFE_LIB=$lib NOISE=$noise bash tools/trace_serial.sh 2>&1 | grep -E 'ransac_hyp|ransac_refine|select_ransac_kernel<1>' | awk '{for(i=1;i<=NF;i++) if($i=="dur") printf "%s ", $(i+1)} END{print ""}')"
  done
done
