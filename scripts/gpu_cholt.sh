cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3; do
MCP_CHOL_DBG=$d timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ct$d -- python $GRAFT_REPO_ROOT/scripts/chol_time.py > /dev/null 2>&1
echo "dbg=$d"; grep -E "k_chol_step|k_chol_back" $GRAFT_REPO_ROOT/gpurun_out/ct$d/*/*kernel_stats.csv | cut -d, -f1-7 | cut -c1-150
done
