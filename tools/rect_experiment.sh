set -e
timeout -k 10 900 python -m pytest tests/test_gpu_adapter.py tests/test_gpu_twostage.py "tests/test_gpu_parity.py::test_unrestricted_scf_energy_parity" -x -q -m gpu > gpurun_out/r02_t7.txt 2>&1 || true
tail -12 gpurun_out/r02_t7.txt
for r in 0 1 0 1; do HELFEM_GEMM_RECT=$r python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rect=$r', d['value'], d['stages_ms'])"; done > gpurun_out/r02_rect.txt 2>&1
cat gpurun_out/r02_rect.txt
HELFEM_GEMM_RECT=1 timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "eig or eigen" > gpurun_out/r02_t8.txt 2>&1 || true
tail -3 gpurun_out/r02_t8.txt
