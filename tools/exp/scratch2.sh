cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for lim in 1000000000 2000000000; do
echo "== HSA_SCRATCH_SINGLE_LIMIT=$lim"
HSA_SCRATCH_SINGLE_LIMIT=$lim python bench.py --cpu-sample 0 2>&1 | tail -1 | cut -c1-120
done
