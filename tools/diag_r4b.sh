cd $GRAFT_REPO_ROOT
echo "--- diag no serialize"; timeout 200 python tools/diag_r4.py 2>&1 | tail -12
echo "--- rulebook bench"; RB_ONLY_SUBM=1 timeout 300 python -X faulthandler tools/rulebook_bench.py 2>&1 | tail -30
echo "--- bench short"; timeout 300 python -X faulthandler bench.py --steps 16 --warmup 8 --no-cpu-baseline --scenes 4 --no-also 2>&1 | tail -40
