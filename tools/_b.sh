python -m pytest tests/test_gpu_conv.py tests/test_gpu_modules.py tests/test_gpu_norm.py tests/test_gpu_int8.py -x -q 2>&1 | tail -3
b() { env $3 python bench.py --config $1 --steps 40 --warmup 10 --no-cpu-baseline $2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('$1 $2 $3', r['ms_per_step'], r.get('eager_ms_per_step'))"; }
b 4 "" X=1; b 4 --key-order X=1; b 4i "" X=1; b 4i --key-order X=1
