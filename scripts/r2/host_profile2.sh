#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
rm -rf /tmp/hp && mkdir -p /tmp/hp
timeout -k 10 500 python -c "
import sys, cProfile, pstats; sys.path.insert(0,'is-dqn_amd')
from experiments.atari.isdqn import run
argv='-en hp_Synthetic -s 1 -dw -f 32 64 64 512 -at cnn -ln -nbi 9 -rbc 20000 -bs 256 -utd 4 -nis 1000 -ed 4000 -tuf 400 -horizon 300 -ne 1 -ntspe 6000 -env synthetic'.split()
pr=cProfile.Profile(); pr.enable(); run(argv, root='/tmp/hp'); pr.disable()
st=pstats.Stats(pr); st.sort_stats('cumulative'); st.print_callees('_flush'); st.print_callees('_graph.py.*run'); st.print_callees('collect_single_sample')
" 2>&1 | grep -v amdgpu.ids | grep -E "^\s+[0-9]|->|called" | cut -c1-170 | head -70
