#!/bin/bash
# Round-3 soak: the trainer end to end on the synthetic environment at the headline widths -- prioritized + n-step with worker
# processes, the impala torso, BatchNorm (cnn and impala), TF-DQN with BatchNorm, LunarLander fc -- a few thousand steps each.
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
rm -rf /tmp/soak && mkdir -p /tmp/soak
t0=$(date +%s)
timeout -k 10 800 python -c "
import sys, time; sys.path.insert(0,'is-dqn_amd')
from experiments.atari.isdqn import run
from experiments.atari.tfdqn import run as run_tf
from experiments.lunar_lander.isdqn import run as run_ll
common='-s 1 -dw -f 32 64 64 512 -ln -rbc 20000 -bs 256 -utd 4 -nis 1000 -ed 4000 -tuf 400 -horizon 300 -ne 2 -env synthetic'.split()
jobs=[('soakA_Synthetic', run, ['-at','cnn','-nbi','9','-ntspe','6000','-per','-n','3','-nenvs','8','-nworkers','2']),
      ('soakB_Synthetic', run, ['-at','impala','-nbi','9','-ntspe','1500']),
      ('soakC_Synthetic', run, ['-at','cnn','-nbi','9','-ntspe','3000','-bn','-a']),
      ('soakD_Synthetic', run, ['-at','impala','-nbi','4','-ntspe','1000','-bn']),
      ('soakE_Synthetic', run_tf, ['-at','cnn','-ntspe','3000','-bn'])]
for name, fn, extra in jobs:
    t=time.time(); g=fn(['-en',name]+common+extra, root='/tmp/soak'); print(name, 'epochs', len(g), 'env steps/s', [round(float(x[0][3]),1) for x in g], 'wall', round(time.time()-t,1), flush=True)
t=time.time(); g=run_ll('-en soakLL -s 1 -dw -f 100 100 -at fc -nbi 1 -rbc 10000 -bs 32 -utd 1 -nis 500 -ed 2000 -tuf 100 -horizon 200 -ne 2 -ntspe 4000 -env synthetic -ln'.split(), root='/tmp/soak'); print('lunar_lander', len(g), round(time.time()-t,1), flush=True)
" 2>&1 | grep -v amdgpu.ids | tail -14
echo "total $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('/tmp/soak/*/exp_output/*/*/episode_returns_and_lengths/1.json')):
    d = json.load(open(f)); print(f.split('exp_output/')[1], 'epochs', len(d['episode_returns']), 'episodes', [len(e) for e in d['episode_returns']])
PY
