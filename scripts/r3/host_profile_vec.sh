#!/bin/bash
# where the vectorised trainer's host time goes (cProfile of a short synthetic-environment run at the headline network)
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
rm -rf /tmp/hpv && mkdir -p /tmp/hpv gpurun_out
timeout -k 10 500 python -c "
import sys, cProfile, pstats; sys.path.insert(0,'is-dqn_amd')
from experiments.atari.isdqn import run
argv='-en hp_Synthetic -s 1 -dw -f 32 64 64 512 -at cnn -ln -nbi 9 -rbc 50000 -bs 256 -utd 4 -nis 2000 -ed 4000 -tuf 8000 -horizon 1000 -ne 1 -ntspe 40000 -env synthetic -nenvs 64 -nworkers 8'.split()
pr=cProfile.Profile(); pr.enable(); run(argv, root='/tmp/hpv'); pr.disable()
st=pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
" 2>&1 | grep -v amdgpu.ids | tail -44 | cut -c1-170 > gpurun_out/host_profile_vec.txt; cat gpurun_out/host_profile_vec.txt
