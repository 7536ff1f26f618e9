#!/bin/bash
{ for seed in 911 4242; do
    MLDSA_SOAK_SECONDS=240 MLDSA_SOAK_SEED=$seed MLDSA_SOAK_MAX_N=400 python -m pytest tests/test_gpu_sign_schedule.py -m gpu -k soak -s -q 2>&1 | grep -E "^soak:|passed|failed|Error" | sed "s/^/seed $seed (small calls, n <= 400): /"
  done
  MLDSA_SOAK_SECONDS=300 MLDSA_SOAK_SEED=31337 python -m pytest tests/test_gpu_sign_schedule.py -m gpu -k soak -s -q 2>&1 | grep -E "^soak:|passed|failed|Error" | sed "s/^/seed 31337 (n <= 70 000): /"
  MLDSA_SOAK_S=120 python -m pytest tests/test_gpu_batcher.py -m gpu -k soak -s -q 2>&1 | grep -E "soak|passed|failed|Error" | sed "s/^/batcher soak 120 s: /"
}
