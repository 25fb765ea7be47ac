#!/bin/bash
# usage (GPU box): bash scripts/profile_all.sh <which>   which = headline | small | c3 | c4
# Runs scripts/profile_round.sh for the named configurations; results under gpurun_out/<tag>/ to be copied
# to profiles/rNN_<tag>_{bench.json,kernel_stats.csv,pmc.json}.
case "$1" in
  headline) bash scripts/profile_round.sh r06_final ;;
  small)    bash scripts/profile_round.sh r06_config1 --config 1 && bash scripts/profile_round.sh r06_config2 --config 2 ;;
  c3)       bash scripts/profile_round.sh r06_config3 --config 3 ;;
  c4)       bash scripts/profile_round.sh r06_config4 --config 4 ;;
  *) echo "headline | small | c3 | c4"; exit 2 ;;
esac
