for v in "" bk32; do
  if [ -n "$v" ]; then export SCTC_LIB_PATH=$PWD/stanford-ctc_amd/libvar_$v.so; else unset SCTC_LIB_PATH; fi
  echo "variant=$v"; python tests/gpu_diag.py gemm 2>&1 | grep "gemm " | head -6
  python bench.py --no-cpu-baseline --no-side 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['phase_ms']['fwd_gemm'], d['phase_ms']['bwd_gemm'])"
done
