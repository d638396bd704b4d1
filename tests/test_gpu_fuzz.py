"""The randomised differential tests (tests/gpu_fuzz*.py) as part of the suite: fixed seeds, a
20-case subset each (the scripts themselves take [n_cases] [seed] for longer hand runs).
  recurrent kernels: the automatically chosen kernel vs the flag kernel and the float64 oracle;
  CTC: vs the C oracle (blanks inside label rows, T < U, infeasible repeats, zero-probability
       labels, peaked rows, 1- and 4-wave lattices, fp32 / fp64);
  GEMM: the C-ABI sctc_gemm_f32 vs a float64 product (all layouts, ragged sizes, split-K)."""
import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_recurrent_kernels(monkeypatch):
    from tests import gpu_fuzz
    monkeypatch.setenv("SCTC_REC_VARIANT", "0")      # the script flips it per net; restore afterwards
    gpu_fuzz.run(20, seed=1)


def test_fuzz_ctc():
    from tests import gpu_fuzz_ctc
    gpu_fuzz_ctc.run(20, seed=2)


def test_fuzz_gemm():
    from tests import gpu_fuzz_gemm
    gpu_fuzz_gemm.run(60, seed=3)
