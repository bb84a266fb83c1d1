for i in 1 2; do for l in libecseg_hip.so libecseg_vnt2.so; do for b in 64 16; do
v=$(ECSEG_HIP_LIB=$GRAFT_REPO_ROOT/ecseg_amd/$l timeout -k 10 200 python bench.py --base $b --images $((1024/b)) --group $((b==64?16:0)) --no-cpu-baseline --no-narrow --no-host-inclusive 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
echo "$l base $b: $v"; done; done; done
