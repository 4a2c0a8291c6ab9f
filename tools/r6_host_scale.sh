#!/bin/bash
# Round 6 (review item 8): the HOST side of an 8-rank run on the 16-core quota of a GPU box, device stages stubbed at the measured
# kernel time (tools/host_scale.py): one rank alone, eight ranks side by side, and eight ranks through rank 0's ordered writer - packed
# blocks (the product's transport since round 6) against the pickled messages of rounds 2-5.
cd ${GRAFT_REPO_ROOT:-/root/repo}
CALL_MS=${1:-48.6}
for rep in 1 2; do
  echo "== one rank alone (rep $rep)";  taskset -c 0-15 python tools/host_scale.py --ranks 1 --reads 20000 --call-ms $CALL_MS 2>&1 | grep -v Gloo | tail -1
  echo "== eight ranks, no merge";      taskset -c 0-15 python tools/host_scale.py --ranks 8 --reads 20000 --call-ms $CALL_MS 2>&1 | grep -v Gloo | tail -1
  echo "== eight ranks, merge, packed"; taskset -c 0-15 python tools/host_scale.py --ranks 8 --reads 160000 --call-ms $CALL_MS --merge 2>&1 | grep -v Gloo | tail -1
  echo "== eight ranks, merge, pickled"; taskset -c 0-15 python tools/host_scale.py --ranks 8 --reads 160000 --call-ms $CALL_MS --merge --pickled 2>&1 | grep -v Gloo | tail -1
done
