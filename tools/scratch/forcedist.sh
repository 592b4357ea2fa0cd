#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for args in "--steps 200" "--force-dist --steps 200" "--force-dist --steps 1000" "--pairs 128 --steps 1000" "--force-dist --steps 1000 --pairs 128" "--force-dist --steps 1000 --pairs 128 --graph off"; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 200 python3 bench.py --cpu-seconds 0 $args > /tmp/fd.json 2> /tmp/fd.err
  rc=$?
  echo "rc=$rc $args"
  if [ $rc -ne 0 ]; then tail -5 /tmp/fd.err; else python3 -c "
import json
j=json.loads(open('/tmp/fd.json').read().strip().splitlines()[-1])
print('   ', j['value'], j['ms_per_step'], j['config']['streams'], j['config']['graph_replay'], j['config']['reduce'], j['parity'])"; fi
done
