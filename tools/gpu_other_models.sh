#!/bin/bash
# the other configurations (ViT-L/14, ViT-B/32, image tower alone, FLAVA cfg 4, CoCa cfg 5, training steps), each against the r05 build when it is
# present (multimodal_amd/lib_base/libmmamd_r05.so) as alternating same-box arms:  bash tools/gpu_other_models.sh [tag]
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; cd $GRAFT_REPO_ROOT
T=${1:-r06}
BASE=multimodal_amd/lib_base/libmmamd_r05.so
run() {  # run <label> <cmd...>
  label=$1; shift
  if [ -f $BASE ]; then echo "{\"arm\": \"r05 build\", \"what\": \"$label\"}"; MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=$BASE timeout 400 "$@" 2>/dev/null | grep '^{' | tail -1; fi
  echo "{\"arm\": \"this build\", \"what\": \"$label\"}"; timeout 400 "$@" 2>/dev/null | grep '^{' | tail -1
}
{
run "CLIP ViT-L/14 B=256 fwd+loss" python tools/clip_fwd_bench.py --model l14 --steps 10
run "CLIP ViT-B/32 B=256 fwd+loss" python tools/clip_fwd_bench.py --model b32 --steps 20
run "CLIP ViT-B/16 image tower alone" python tools/clip_fwd_bench.py --model b16 --vision-only --steps 20
run "FLAVA cfg 4 B=128 fwd + pre-training loss" python tools/flava_bench.py
run "FLAVA cfg 4, attentions opted out" python tools/flava_bench.py --no-attentions
run "CoCa L/14 B=128 fwd + losses" python tools/coca_bench.py
run "CLIP ViT-B/16 training step" python tools/train_bench.py
run "FLAVA training step" python tools/flava_bench.py --train --steps 6
} > $O/${T}_other_models.jsonl
cut -c1-330 $O/${T}_other_models.jsonl
