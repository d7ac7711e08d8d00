#!/bin/bash
# Start W ranks of examples/shard_client.cpp, one per GPU:  examples/run_shard_client.sh W [points] [queries]
W=${1:-1}; shift
DIR=$(cd "$(dirname "$0")/.." && pwd)
TOKEN=$(mktemp -u /tmp/pct_shard_token.XXXXXX)
export HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for ((r = 0; r < W; r++)); do
  "$DIR/pointcloudtraj_amd/lib/shard_client" $r $W $TOKEN "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
rm -f $TOKEN
exit $rc
