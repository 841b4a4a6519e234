#!/bin/bash
# soak: the drop-in trainer end to end on the synthetic environment at the headline network, three flavours
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
rm -rf /tmp/soak && mkdir -p /tmp/soak
common="-s 1 -dw -f 32 64 64 512 -at cnn -ln -nbi 9 -rbc 20000 -bs 256 -utd 4 -nis 1000 -ed 4000 -tuf 400 -horizon 300 -ne 2 -ntspe 3000 -env synthetic"
t0=$(date +%s)
timeout -k 10 500 python -c "
import sys; sys.path.insert(0,'is-dqn_amd')
from experiments.atari.isdqn import run
import time
for name, extra in (('soakA_Synthetic', []), ('soakB_Synthetic', ['-per','-n','3','-nenvs','4']), ('soakC_Synthetic', ['-a','-hd','1.0'])):
    t=time.time(); g=run(['-en',name]+'$common'.split()+extra, root='/tmp/soak'); print(name, 'epochs', len(g), 'env steps/s', [round(float(x[0][3]),1) for x in g], 'wall', round(time.time()-t,1), flush=True)
from experiments.atari.dqn import run as run_dqn
t=time.time(); g=run_dqn(['-en','soakD_Synthetic']+'$common'.replace('-nbi 9','').split(), root='/tmp/soak'); print('dqn', len(g), round(time.time()-t,1))
" 2>&1 | grep -v amdgpu.ids | tail -12
echo "total $(( $(date +%s) - t0 )) s"; ls /tmp/soak/atari/exp_output/
