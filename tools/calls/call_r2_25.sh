cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in 3072 4096 8192 16384 65536; do
  tools/calls/call_r2_ab_bench.sh "hip lat" --steps 300 --batch $B 2>&1 | sed "s/^/B=$B /" | sort | awk '{k=$1" "$2; s[k]=s[k]" "$8} END{for(k in s) print k, s[k]}' | sort
done
