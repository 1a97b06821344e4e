#!/bin/bash
# end-to-end check of the reference launcher surface on a synthetic Yoochoose-1/64-shaped dataset:
# DataLoader workers + native collate + TrainRunner (hipGraph replay vs eager).  usage: tools/e2e_launcher.sh [workers]
W=${1:-4}
D=/tmp/synth_yc
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
import bench
rng = np.random.default_rng(123)
V = 37484
tr = bench.synth_sessions(60000, V, 6.2, 20, rng)
te = bench.synth_sessions(2000, V, 6.2, 20, rng)
os.makedirs('/tmp/synth_yc', exist_ok=True)
for name, ss in (('train.txt', tr), ('test.txt', te)):
    with open('/tmp/synth_yc/' + name, 'w') as f:
        for s in ss:
            f.write(','.join(str(int(x)) for x in s) + '\n')
open('/tmp/synth_yc/num_items.txt', 'w').write(str(V))
print('sessions', len(tr), 'clicks', sum(len(s) for s in tr))
PY
cd /root/repo/src
for mode in "--precision bf16" "--precision bf16 --no-graph" "--precision fp32"; do
  echo "== MSGIFSR $mode workers=$W"
  python scripts/main_msgifsr.py --dataset-dir $D --epochs 1 --batch-size 512 --num-workers $W --log-interval 100 $mode 2>&1 | grep -E "Batch [0-9]+00:|training steps|Epoch|MRR@20" | head -12
done
