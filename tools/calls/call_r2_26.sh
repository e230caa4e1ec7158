cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for p in 1 0 1 0; do
  echo "== GTOP_POLL_COMPLETION=$p"
  GTOP_POLL_COMPLETION=$p timeout -k 5 200 python3 tools/host_api_rate.py 1 2>&1 | grep -v amdgpu.ids | grep "B=1\|16 threads\|64 threads" | cut -c1-230
done
